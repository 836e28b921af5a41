#!/bin/bash
# Dynamic instruction mix of the hot kernels of the bench step (per-class SQ_INSTS_* counters), two --pmc passes.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3/mix; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_INSTS_[A-Z0-9_]*" $OUT/counters.txt | sort -u > $OUT/sq_insts.txt
A="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
B="SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"
timeout -k 10 240 rocprofv3 --pmc $A --output-format csv -d $OUT/a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-config5 --no-stream > $OUT/a.log 2>&1; echo "a rc=$?"
timeout -k 10 240 rocprofv3 --pmc $B --output-format csv -d $OUT/b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power --no-config5 --no-stream > $OUT/b.log 2>&1; echo "b rc=$?"
python3 - "$OUT" <<'P'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(("demod_kernel", "corr_kernel")):
            acc[(k, int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key, {c: sum(v) / len(v) for c, v in sorted(cs.items())})
P
