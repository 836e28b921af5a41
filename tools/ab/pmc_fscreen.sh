#!/bin/bash
# SQ counters of the frames-sync kernels (all-fp64 corr_kernel against the screened corr_screen_kernel) on the headline batch.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4/pmc_fs; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $R/tools/ab/time_screened_sync.py > $OUT/sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- python3 $R/tools/ab/time_screened_sync.py > $OUT/lds.log 2>&1; echo "lds rc=$?"
python3 - "$OUT" <<'P'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(("corr_kernel", "corr_screen")):
            acc[(k, int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    print(key)
    for c, v in m.items(): print("   %-24s %.4g" % (c, v))
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        print("   shares of wave cycles: VALU %.2f LDS %.2f wait_any %.2f wait_inst %.2f" % (m["SQ_ACTIVE_INST_VALU"] / w, m["SQ_ACTIVE_INST_LDS"] / w, m["SQ_WAIT_ANY"] / w, m["SQ_WAIT_INST_ANY"] / w))
P
