#!/usr/bin/env python3
"""Timing-only driver for ablation builds of the stream sync (results are NOT checked: such builds compute wrong values on
purpose; the cells handed to the re-evaluation come from the screen and do not change)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import time, torch
spec = importlib.util.spec_from_file_location("gf3_config3", os.path.join(ROOT, "tools", "config3.py"))
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
eng, cfg, channel = tool.make_engine()
r, payload = tool.make_stream(eng, channel, 4096)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
walls = []
pause = float(os.environ.get("GF3_PAUSE_MS", "0")) * 1e-3        # idle time between calls (does the clock governor matter?)
for i in range(23):
    if pause: time.sleep(pause)
    t0 = time.perf_counter(); ev[0].record(); eng.sync_stream(r, 8192); t1 = time.perf_counter(); ev[1].record(); torch.cuda.synchronize()
    if i >= 3: ts.append(ev[0].elapsed_time(ev[1])); walls.append((t1 - t0) * 1e3)
ts.sort()
walls.sort()
print(os.environ.get("GF3_LIB", "in-tree"), "sync median %.3f ms by events, %.3f ms host wall time of the call (unchecked)" % (ts[len(ts) // 2], walls[len(walls) // 2]))
