#!/usr/bin/env python3
"""Where the time of Engine.receive_host goes on the config-3 stream from pinned memory: plain copies of several piece sizes,
then the call at several piece sizes (GF3_LIB selects the build)."""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("gf3_config3", os.path.join(ROOT, "tools", "config3.py"))
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
FRAMES = int(os.environ.get("GF3_H2D_FRAMES", "4096"))       # 28000: a 2.19 G-sample stream (lag indices past 2^31), 8.8 GB of f32
eng, cfg, channel = tool.make_engine()
r, payload = tool.make_stream(eng, channel, FRAMES)
host = torch.empty(r.numel(), dtype=r.dtype).pin_memory(); host.copy_(r); torch.cuda.synchronize()
n = host.numel()
dst = torch.empty_like(r)
for piece in (1 << 23, 1 << 25, 1 << 27, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for lo in range(0, n, piece):
        dst[lo:lo + piece].copy_(host[lo:lo + piece], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("plain copy in pieces of %10d samples: %.2f ms = %.1f GB/s" % (piece, dt * 1e3, n * 4 / dt / 1e9), flush=True)
for chunk in (1 << 24, 1 << 25, 1 << 26, 1 << 27):
    ts = []
    for _ in range(4):
        t = time.perf_counter(); res = eng.receive_host(host, chunk_samples=chunk); ts.append(time.perf_counter() - t)
    t = float(np.median(ts[1:]))
    i = res["info"]
    print("receive_host chunk %10d: %.2f ms = %.1f GB/s, %.2f G samples/s, pieces %d (last call: setup %.2f, pieces %.2f, total %.2f ms)" % (
        chunk, t * 1e3, n * 4 / t / 1e9, n / t / 1e9, i["chunks"], i["setup_seconds"] * 1e3, i["pieces_seconds"] * 1e3, i["seconds"] * 1e3), flush=True)
pageable = host.numpy().copy() if FRAMES <= 4096 else None
if FRAMES <= 4096:                                          # ... and the default for pageable memory (large streams: pieces of 128 MiB and more handed to the runtime; small ones staged)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); res3 = eng.receive_host(pageable, chunk_samples=1 << 25); ts.append(time.perf_counter() - t)
    t = float(np.median(ts[1:]))
    print("receive_host from PAGEABLE memory, default (%s): %.2f ms = %.1f GB/s; peaks equal: %s" % (res3["info"]["source"], t * 1e3, n * 4 / t / 1e9, bool(torch.equal(res3["peaks"], res["peaks"]))), flush=True)
one = eng.sync_stream(r)
print("peaks equal the one-shot path:", bool(torch.equal(one, res["peaks"])), " bits equal:", bool(torch.equal(eng.demod_frames(r, (one + 2)[:-1])["bits"], res["bits"])),
      " detections", int(one.numel()), res["info"])
