#!/bin/bash
# Run ON THE GPU BOX from the repo root (delete the local gpurun_out/round first: gpurun MERGES scratch output, and the
# collector averages every CSV it finds there): bench line, rocprofv3 kernel stats of the same command, HBM-traffic counters
# (separate --pmc passes, as the microarch guide prescribes) and one SQ pass (issue / wait / LDS shares of the hot kernels).
set -o pipefail
R=$PWD; OUT=$R/gpurun_out/round; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu --no-power --no-h2d > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-h2d > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-h2d > $OUT/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-config5 --no-by-n --no-h2d > $OUT/sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/grbm -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-config5 --no-by-n --no-stream > $OUT/grbm.log 2>&1; echo "grbm rc=$?"
# dynamic instruction mix of the two hot kernels of the step (per-class SQ_INSTS_* counters, two passes): roofline.power's model
A="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
B="SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"
rocprofv3 --pmc $A --output-format csv -d $OUT/mixa -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-config5 --no-by-n --no-stream --no-pcm16 --no-final-system-test > $OUT/mixa.log 2>&1; echo "mixa rc=$?"
rocprofv3 --pmc $B --output-format csv -d $OUT/mixb -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-config5 --no-by-n --no-stream --no-pcm16 --no-final-system-test > $OUT/mixb.log 2>&1; echo "mixb rc=$?"
cd $R; ls $OUT
