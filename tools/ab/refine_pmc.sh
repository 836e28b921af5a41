#!/bin/bash
# Where scr_refine_kernel's cycles go on the config-3 stream: kernel-trace durations of the build under GF3_LIB (default in-tree),
# then three --pmc passes (wave / LDS / vector-memory counters) restricted to that kernel.   tools/ab/refine_pmc.sh [tag]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
R=$GRAFT_REPO_ROOT; TAG=${1:-prod}; OUT=$R/gpurun_out/r3/refpmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INST_LEVEL_LDS"
C="TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VALU"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/ab/time_config3.py > $OUT/t.log 2>&1 || { echo "trace failed"; tail -5 $OUT/t.log; exit 1; }
for p in A B C; do
  timeout -k 10 200 rocprofv3 --pmc ${!p} --output-format csv -d $OUT/$p -- python3 $R/tools/ab/time_config3.py > $OUT/$p.log 2>&1 || { echo "pass $p failed"; tail -5 $OUT/$p.log; exit 1; }
done
python3 - "$OUT" <<'P'
import collections, csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "t", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "scr_re" in row["Name"] or "scr_ring" in row["Name"]:
            print(row["Name"][:60], "calls", row["Calls"], "avg ns", row["AverageNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "[ABC]", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("scr_refine"):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
P
