#!/usr/bin/env python3
"""Time the fp32 screening kernel alone (gf3_debug_stream_screen) on the config-3 stream; GF3_LIB selects the build."""
import importlib.util, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("c3", os.path.join(ROOT, "tools", "config3.py"))
c3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c3)
eng, cfg, channel = c3.make_engine()
r, payload = c3.make_stream(eng, channel, 4096)
ts = []
for i in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = eng.debug_stream_screen(r); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(json.dumps({"lib": os.environ.get("GF3_LIB", "in-tree"), "screen_ms": ts}))
