#!/bin/bash
# kernel trace of the stream-sync kernels on the config-3 stream (GF3_LIB selects the build)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/c3trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/config3.py > $OUT/trace.log 2>&1; echo "trace rc=$?"
tail -1 $OUT/trace.log | cut -c1-300
grep -h "scr_\|pk_" $(find $OUT/trace -name "*kernel_stats.csv") | cut -c1-150
