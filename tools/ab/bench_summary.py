"""One line per bench JSON: the figures that are compared between runs.  python tools/ab/bench_summary.py a.json [b.json ...]"""
import json, sys
for p in sys.argv[1:]:
    d = json.load(open(p))
    f = d.get("final_system_test", {})
    print(p, "value %.1f G/s" % (d["value"] / 1e9), "step %.3f ms" % d["ms_per_step"], "demod %.3f ms frac %.4f" % (d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]),
          "sync %.3f ms" % d.get("roofline_sync", {}).get("avg_launch_ms", float("nan")), "| fst %.3f ms sha %s ber %s" % (f.get("ms_per_call", float("nan")), f.get("bits_sha256_equal_reference"), f.get("ber_string_equal_reference")),
          "| bit_errors", d.get("bit_errors"), "sync_exact", d.get("sync_exact"))
