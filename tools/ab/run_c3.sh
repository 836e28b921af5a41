#!/bin/bash
# timing helpers for the screening kernel on the config-3 stream (in-tree library unless GF3_LIB is set)
cd $GRAFT_REPO_ROOT
python tools/ab/time_screen.py 2>&1 | tail -1
python tools/config3.py 2>&1 | tail -1 | cut -c1-160
