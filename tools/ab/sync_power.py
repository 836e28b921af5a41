#!/usr/bin/env python3
"""Diagnostic: the config-3 stream sync repeated back to back for a few seconds -- time per call, package power and the
shader clock the chip settles at (GF3_LIB selects the build).  Is the call bound by its energy, like the fp64 kernels?"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("gf3_config3", os.path.join(ROOT, "tools", "config3.py"))
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
spec = importlib.util.spec_from_file_location("gf3_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
eng, cfg, channel = tool.make_engine()
r, payload = tool.make_stream(eng, channel, 4096)
for _ in range(3): eng.sync_stream(r, 8192)
torch.cuda.synchronize()
secs = float(os.environ.get("GF3_SECONDS", "3"))
ps = bench.PowerSampler(0); n = 0
with ps:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(20): eng.sync_stream(r, 8192)
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
w = float(np.median(ps.samples[len(ps.samples) // 2:])) if ps.samples else float("nan")
print(os.environ.get("GF3_LIB", "in-tree"), "calls %d  %.4f ms per call (wall, back to back)  package %.0f W  => %.3f J per call" % (n, dt / n * 1e3, w, w * dt / n))
