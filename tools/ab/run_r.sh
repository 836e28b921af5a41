#!/bin/bash
cd $GRAFT_REPO_ROOT
export GF3_LIB=$PWD/tools/ab/dev.so
for R in 110 164 200 328; do GF3_SCR_R=$R timeout -k 10 120 python tools/ab/time_screen.py 2 2>&1 | tail -1 || exit 1; done
GF3_SCR_R=164 bash tools/ab/pmc_c3.sh
