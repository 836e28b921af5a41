#!/bin/bash
# same-box A/B of library builds on the headline step: tools/ab/ab_bench.sh OUTDIR lib1.so lib2.so ... (each run twice, interleaved)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
OUT="$1"; shift
mkdir -p "$OUT"
for pass in 1 2; do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    GF3_LIB="$lib" timeout -k 10 300 python bench.py --no-cpu --no-config5 --no-stream --no-power > "$OUT/$name.$pass.json" 2> "$OUT/$name.$pass.err" || { echo "FAILED $name pass $pass"; tail -5 "$OUT/$name.$pass.err"; exit 1; }
    python - "$OUT/$name.$pass.json" "$name.$pass" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value %.1f G/s" % (d["value"] / 1e9), "sync %.3f ms" % d["roofline_sync"]["avg_launch_ms"], "demod %.3f ms" % d["roofline"]["avg_launch_ms"],
      "bit_errors", d["bit_errors"], "sync_exact", d["sync_exact"])
P
  done
done
