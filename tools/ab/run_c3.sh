#!/bin/bash
# timing helpers for the screening kernel on the config-3 stream (in-tree library unless GF3_LIB is set)
cd $GRAFT_REPO_ROOT
python tools/ab/time_screen.py 2>&1 | tail -1
python tools/config3.py 2>&1 | tail -1 | cut -c1-160
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3fetch -- python3 $GRAFT_REPO_ROOT/tools/config3.py > $GRAFT_REPO_ROOT/gpurun_out/c3fetch.log 2>&1; echo rc=$?
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3write -- python3 $GRAFT_REPO_ROOT/tools/config3.py > $GRAFT_REPO_ROOT/gpurun_out/c3write.log 2>&1; echo rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3trace -- python3 $GRAFT_REPO_ROOT/tools/config3.py > $GRAFT_REPO_ROOT/gpurun_out/c3trace.log 2>&1; echo rc=$?
