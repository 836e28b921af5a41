#!/bin/bash
cd $GRAFT_REPO_ROOT
GF3_LIB=$PWD/tools/ab/scr_v0.so python tools/config3.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && GF3_LIB=$GRAFT_REPO_ROOT/tools/ab/scr_v0.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3prof7 -- python3 $GRAFT_REPO_ROOT/tools/config3.py > $GRAFT_REPO_ROOT/gpurun_out/c3prof7.log 2>&1; echo rc=$?
