#!/bin/bash
# SQ / LDS counters of the stream-sync kernels on the config-3 stream (GF3_LIB selects the build)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/c3pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $R/tools/config3.py > $OUT/sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/grbm -- python3 $R/tools/config3.py > $OUT/grbm.log 2>&1; echo "grbm rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/config3.py > $OUT/trace.log 2>&1; echo "trace rc=$?"
python3 $R/tools/ab/pmc_sum.py $OUT/sq scr_ ; python3 $R/tools/ab/pmc_sum.py $OUT/grbm scr_
grep -h "scr_\|pk_" $(find $OUT/trace -name "*kernel_stats.csv") | cut -c1-150
