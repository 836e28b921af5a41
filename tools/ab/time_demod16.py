#!/usr/bin/env python3
"""Same-box timing of the 16-QAM table mode of the fused kernel (bench.py's roofline_demod_16qam leg alone; GF3_LIB selects the build)."""
import importlib.util, json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("gf3_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
args = types.SimpleNamespace(frames=65536, stride=78720, window=320)
r = bench.demod_16qam_roofline(torch.device("cuda", 0), args)["roofline_demod_16qam"]
print(os.environ.get("GF3_LIB", "in-tree"), "16-QAM demod %.3f ms  frac %.4f  payload_recovered %s" % (r["avg_launch_ms"], r["frac"], r["payload_recovered"]))
