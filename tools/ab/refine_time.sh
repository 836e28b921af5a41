#!/bin/bash
# Kernel-trace duration of the stream-sync kernels on the config-3 stream, one library after another.
#   tools/ab/refine_time.sh build/a.so build/b.so ...     ("-" = the in-tree library)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "$lib" .so); OUT=$R/gpurun_out/r3/reftime_$tag; mkdir -p $OUT
  if [ "$lib" = "-" ]; then unset GF3_LIB; else export GF3_LIB=$R/$lib; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/ab/${GF3_AB_DRIVER:-time_config3.py} > $OUT/log.txt 2>&1 || { echo "$tag failed"; tail -5 $OUT/log.txt; exit 1; }
  tail -1 $OUT/log.txt
  python3 - "$OUT" "$tag" <<'P'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Name"].startswith(("void scr_re", "void scr_ring")):
            print("   %-8s %-44s calls %3s  avg %9.1f us" % (sys.argv[2], row["Name"][:44], row["Calls"], float(row["AverageNs"]) / 1e3))
P
done
