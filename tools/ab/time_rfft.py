#!/usr/bin/env python3
"""rfft_batch at N=4096 (dev builds serve this size) over 2^28 samples with preallocated output; GF3_RFFT_PER_WG = transforms per workgroup."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
total = 1 << 28
x = torch.randn(total, dtype=torch.float32, device="cuda")
K = N // 2 - 1
pts, bt = qpsk_table()
cfg = RxConfig(N=N, CP=0, P=1, D=1, data_bins=np.arange(1, K), const_points=pts, const_bits=bt,
               known_bits=np.zeros(2 * K, np.uint8), in_dtype=torch.float32, fit_lo=10, fit_hi=100)
eng = Engine(cfg)
n_sym = total // N
off = torch.arange(n_sym, dtype=torch.int64, device="cuda") * N
out = torch.empty((n_sym, N // 2 + 1), dtype=torch.complex128, device="cuda")
eng.rfft_batch(x, off, out=out)
ref = np.fft.rfft(x[: 8 * N].cpu().numpy().astype(np.float64).reshape(8, N))
err = np.abs(out[:8].cpu().numpy() - ref).max() / np.abs(ref).max()
ref2 = np.fft.rfft(x[-3 * N:].cpu().numpy().astype(np.float64).reshape(3, N))
err2 = np.abs(out[-3:].cpu().numpy() - ref2).max() / np.abs(ref2).max()
ts = []
for i in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.rfft_batch(x, off, out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
t = sorted(ts)[len(ts) // 2] * 1e-3
by = n_sym * (4 * N + 16 * (N // 2 + 1))
print(json.dumps({"per_wg": os.environ.get("GF3_RFFT_PER_WG"), "N": N, "ms": round(t * 1e3, 4), "frac": round(by / t / 8e12, 4), "err": float(max(err, err2))}), flush=True)
