#!/bin/bash
# L1 (TCP) and L2 (TCC) request counters of the frames-sync kernels on the headline batch: how much of corr_kernel's chirp-spectrum
# traffic (229 KB per window, the same 229 KB for every window) is served by the CU's L1 today.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4/pmc_cc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcp -- python3 $R/tools/ab/time_screened_sync.py > $OUT/tcp.log 2>&1; echo "tcp rc=$?"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $R/tools/ab/time_screened_sync.py > $OUT/tcc.log 2>&1; echo "tcc rc=$?"
python3 - "$OUT" <<'P'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(("corr_kernel", "corr_screen")):
            acc[(k, int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key)
    for c, v in sorted(cs.items()): print("   %-32s %.4g" % (c, sum(v) / len(v)))
P
