#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in b3 b4; do
  GF3_LIB=$PWD/tools/ab/scr_$v.so python tools/ab/time_screen.py 2>&1 | tail -1
  GF3_LIB=$PWD/tools/ab/scr_$v.so python tools/config3.py 2>&1 | tail -1 | cut -c1-200
  GF3_LIB=$PWD/tools/ab/scr_$v.so timeout -k 10 300 python -m pytest tests -m gpu -q -k "screen_error or screened_equals or falls_back" 2>&1 | tail -3
done
