#!/usr/bin/env python3
"""Time the fp32 screening kernel alone (gf3_debug_stream_screen) and the whole gf3_sync_stream on the config-3 stream;
GF3_LIB selects the build, GF3_SCR_R the ring kernel's blocks per workgroup, argv[1:] the stream modes to time."""
import importlib.util, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("c3", os.path.join(ROOT, "tools", "config3.py"))
c3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c3)
eng, cfg, channel = c3.make_engine()
r, payload = c3.make_stream(eng, channel, 4096)
def ev(f, reps):
    ts = []
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = f(); b.record(); torch.cuda.synchronize()
        ts.append(round(a.elapsed_time(b), 3))
    return sorted(ts)[:3]
for mode in [int(m) for m in sys.argv[1:]] or [2]:
    eng.sync_stream_mode(mode)
    print(json.dumps({"R": os.environ.get("GF3_SCR_R"), "mode": mode, "screen_ms": ev(lambda: eng.debug_stream_screen(r), 6),
                      "sync_ms": ev(lambda: eng.sync_stream(r), 6)}), flush=True)
