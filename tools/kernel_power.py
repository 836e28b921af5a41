#!/usr/bin/env python3
"""Diagnostic: sustained package power, launch time and energy per packet of each hot kernel alone (config 2 batch)."""
import argparse, importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("gf3_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=65536); ap.add_argument("--stride", type=int, default=78720)
ap.add_argument("--window", type=int, default=320)
args = ap.parse_args()
eng, cfg, big, payload, gaps = bench.build_workload(args, 0)
F = args.frames
starts = eng.sync_frames(big, F, args.stride, bench.WIN_LO, bench.WIN_LO + args.window)
bits = torch.empty((F, eng.bytes_per_frame), dtype=torch.uint8, device="cuda")
out = {}
for name, fn in (("corr_kernel", lambda: eng.sync_frames(big, F, args.stride, bench.WIN_LO, bench.WIN_LO + args.window, out_starts=starts)),
                 ("demod_kernel", lambda: eng.demod_frames(big, starts, out_bits=bits))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ps = bench.PowerSampler(0)
    n = 0
    with ps:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 2.5:
            for _ in range(10): fn()
            torch.cuda.synchronize(); n += 10
        dt = time.perf_counter() - t0
    w = float(np.median(ps.samples[len(ps.samples) // 2:])) if ps.samples else None
    out[name] = {"ms_per_launch": dt / n * 1e3, "package_power_w": w, "uJ_per_packet": (w * dt / n / F * 1e6) if w else None}
print(json.dumps(out))
