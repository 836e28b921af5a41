#!/usr/bin/env python3
"""Kernel-level timing of gf3_demod_frames / gf3_sync_frames for several packet shapes
(separates the fixed, per-pilot-symbol and per-data-symbol cost of the fused kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table, square_qam_table

def run(N, CP, P, D, mu, F, reps=5, want=()):
    K = N // 2 - 1
    pts, bt = qpsk_table() if mu == 2 else square_qam_table(mu)
    known = np.unpackbits(np.load(os.path.join(os.path.dirname(__file__), "..", "gf3_audio_modem_amd", "data", "known_bits.npz"))["packed"])
    known = np.tile(known, -(-K * mu // len(known)))
    cfg = RxConfig(N=N, CP=CP, P=P, D=D, data_bins=np.arange(1, K), const_points=pts, const_bits=bt,
                   known_bits=known, in_dtype=torch.float32, max_window=320,
                   fit_lo=min(500, K // 2), fit_hi=min(1000, K))
    eng = Engine(cfg)
    stride = ((300 + cfg.frame_len + 63) // 64) * 64
    gen = torch.Generator(device="cuda").manual_seed(1)
    payload = torch.randint(0, 256, (F, eng.bytes_per_frame), dtype=torch.uint8, device="cuda", generator=gen)
    gaps = torch.randint(0, 300, (F,), dtype=torch.int64, device="cuda", generator=gen)
    filler = np.zeros(K, dtype=complex); filler[K - 1] = pts[0]
    big = eng.tx_frames(payload, filler, stride=stride, gaps=gaps, out_dtype=torch.float32)
    starts = eng.sync_frames(big, F, stride, -8, 312)
    bits = torch.empty((F, eng.bytes_per_frame), dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ts, td = [], []
    for r in range(reps + 2):
        ev[0].record(); eng.sync_frames(big, F, stride, -8, 312); ev[1].record()
        eng.demod_frames(big, starts, out_bits=bits, want=want) if not want else eng.demod_frames(big, starts, want=want)
        ev[2].record(); torch.cuda.synchronize()
        if r >= 2: ts.append(ev[0].elapsed_time(ev[1])); td.append(ev[1].elapsed_time(ev[2]))
    ok = bool(torch.equal(bits, payload)) if not want else None
    print(f"N={N} P={P} D={D} mu={mu} F={F} want={want}: sync {np.median(ts)*1e3/F:.3f} us/frame, demod {np.median(td)*1e3/F:.3f} us/frame, "
          f"bits_ok={ok}", flush=True)
    return np.median(td) * 1e3 / F

if __name__ == "__main__":
    F = 16384
    a = run(4096, 512, 2, 8, 2, F)
    b = run(4096, 512, 2, 24, 2, F)
    c = run(4096, 512, 6, 8, 2, F)
    print(f"per data symbol {(b - a) / 16:.4f} us, per pilot symbol {(c - a) / 8:.4f} us, fixed {a - 8 * (b - a) / 16 - 4 * (c - a) / 8:.4f} us (chip-wide per frame)")
    run(4096, 512, 2, 8, 4, F)
    run(4096, 512, 2, 8, 2, 4096, want=("eq",))
    run(1024, 128, 2, 8, 2, F)
    run(2048, 256, 2, 8, 2, F)
    run(8192, 1024, 2, 8, 2, 4096)
