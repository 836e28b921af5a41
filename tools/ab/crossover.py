#!/usr/bin/env python3
"""Where the screened stream sync overtakes the all-fp64 one: config-3 geometry, streams of F packets."""
import importlib.util, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("c3", os.path.join(ROOT, "tools", "config3.py"))
c3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c3)
eng, cfg, channel = c3.make_engine()
def ev(f, reps=8):
    ts = []
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return round(sorted(ts)[len(ts) // 2], 4)
for F in (4, 8, 16, 32, 64, 128, 256):
    r, _ = c3.make_stream(eng, channel, F)
    out = {"F": F, "n": r.numel()}
    for mode in (2, 1):
        eng.sync_stream_mode(mode)
        out["mode%d_ms" % mode] = ev(lambda: eng.sync_stream(r))
    print(json.dumps(out), flush=True)
