#!/usr/bin/env python3
"""Same-box timing of the config-3 stream sync (GF3_LIB selects the build): median of 20 after 3 warm-ups, peaks checked
against the all-fp64 evaluation."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("gf3_config3", os.path.join(ROOT, "tools", "config3.py"))
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
eng, cfg, channel = tool.make_engine()
r, payload = tool.make_stream(eng, channel, 4096)
res, _, _ = tool.measure(eng, cfg, r, payload, reps=20, warm=3, fp64_reps=3)
print(os.environ.get("GF3_LIB", "in-tree"), "sync %.3f ms  fp64 path %.2f ms  demod %.3f ms  offsets exact %s  %s" % (
    res["sync_stream_s"] * 1e3, res["sync_stream_fp64_path_s"] * 1e3, res["demod_s"] * 1e3, res["sync_offsets_as_expected_plus1"], res["sync_path"]))
