#!/usr/bin/env python3
"""Quick look at BASELINE config 5 on a CACHE-RESIDENT working set (201 MB per N fits the 256 MB Infinity Cache, so the
GB/s printed here are not HBM figures: the roofline numbers come from bench.py's config-5 leg, 2^28 samples per N,
`roofline_rfft` / `roofline_soft_demap`).  Batched real FFT for N in {1024, 2048, 4096, 8192} (2^24 real samples
each, RandomState(5).randn, f32 storage) + 64-QAM soft demapping of 2^24 symbols; reports
kernel time and achieved algorithmic HBM GB/s (B_in*N in + 16*(N/2+1) out per
transform; 16 B in + 4*mu B out per symbol).  Also the known-byte-count workload used to
calibrate the FETCH_SIZE/WRITE_SIZE counters (profiles/)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table, square_qam_table

def ev_time(fn, reps=5):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for r in range(reps + 2):
        e[0].record(); fn(); e[1].record(); torch.cuda.synchronize()
        if r >= 2: ts.append(e[0].elapsed_time(e[1]) * 1e-3)
    return float(np.median(ts))

out = {}
total = 1 << 24
x = torch.from_numpy(np.random.RandomState(5).randn(total).astype(np.float32)).cuda()
for N in (1024, 2048, 4096, 8192):
    K = N // 2 - 1
    pts, bt = qpsk_table()
    cfg = RxConfig(N=N, CP=0, P=1, D=1, data_bins=np.arange(1, K), const_points=pts, const_bits=bt,
                   known_bits=np.zeros(2 * K, np.uint8), in_dtype=torch.float32, fit_lo=10, fit_hi=100)
    eng = Engine(cfg)
    n_sym = total // N
    off = torch.arange(n_sym, dtype=torch.int64, device="cuda") * N
    X = eng.rfft_batch(x, off)
    ref = np.fft.rfft(x[: 4 * N].cpu().numpy().astype(np.float64).reshape(4, N))
    err = np.abs(X[:4].cpu().numpy() - ref).max() / np.abs(ref).max()
    t = ev_time(lambda: eng.rfft_batch(x, off))
    by = n_sym * (4 * N + 16 * (N // 2 + 1))
    out[f"rfft_N{N}"] = {"n_transforms": n_sym, "ms": t * 1e3, "algorithmic_bytes": by, "GBps": by / t / 1e9,
                         "frac_of_8TBps": by / t / 8e12, "max_rel_err_vs_numpy": float(err)}
    print(f"rfft N={N}: {t*1e3:.3f} ms, {by/t/1e9:.0f} GB/s algorithmic ({by/t/8e12:.1%} of 8 TB/s), rel err {err:.1e}", flush=True)
    del X
pts, bt = square_qam_table(6)
cfg = RxConfig(N=1024, CP=0, P=1, D=1, data_bins=np.arange(1, 511), const_points=pts, const_bits=bt,
               known_bits=np.zeros(511 * 6, np.uint8), in_dtype=torch.float32, fit_lo=10, fit_hi=100)
eng = Engine(cfg)
rs = np.random.RandomState(6)
sym = torch.from_numpy(pts[rs.randint(0, 64, total)] + 0.08 * (rs.randn(total) + 1j * rs.randn(total))).cuda()
llr = eng.soft_demap(sym, 0.0128)
hard, _ = eng.demap_hard(sym)
assert torch.equal((llr < 0).to(torch.uint8), hard), "sign(LLR) must equal the hard decision"
t = ev_time(lambda: eng.soft_demap(sym, 0.0128))
by = total * (16 + 4 * 6)
out["soft_demap_64qam"] = {"n_symbols": total, "ms": t * 1e3, "algorithmic_bytes": by, "GBps": by / t / 1e9,
                           "frac_of_8TBps": by / t / 8e12, "sign_equals_hard": True}
print(f"soft demap 64-QAM: {t*1e3:.3f} ms, {by/t/1e9:.0f} GB/s algorithmic ({by/t/8e12:.1%})", flush=True)
print(json.dumps(out))
