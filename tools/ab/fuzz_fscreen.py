#!/usr/bin/env python3
"""Random geometries / storage / noise / thresholds / interferers / window widths through the frames-mode sync: the screened call
(gf3_sync_frames_ex mode 1: fp32 screen with a proven bound + the fp64 kernel on unresolved windows) against the all-fp64 call
-- the indices must be identical -- and the screen's bound against fp64 dot products on the host for a sample of windows.
argv[1] = cases, argv[2] = seed."""
import os, sys, time, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import gf3_oracle as orc
from tests.test_properties import _params
from gf3_audio_modem_amd import Engine, RxConfig
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
bad, unresolved, windows, t0, worst = 0, 0, 0, time.time(), 0.0
for case in range(ncase):
    N = int(rs.choice([1024, 2048, 4096, 8192])); F = int(rs.randint(1, 9)); cp = float(rs.choice([1 / 32, 1 / 8, 1 / 4]))
    storage = str(rs.choice(["float64", "float32", "int16", "uint8"])); snr = float(rs.choice([60.0, 20.0, 6.0, 0.0, -6.0, -12.0]))
    thresh = float(rs.choice([0.4, 0.4, 0.25, 0.6, 0.9])); wmax = int(rs.choice([64, 320, 512])); W = int(rs.randint(3, wmax + 1))
    p = dataclasses.replace(_params(N, cp, 1, 2, 2, 0.0, 0.0), thresh=thresh)
    dt = getattr(torch, storage)
    cfg = RxConfig(N=p.N, CP=p.CP, P=p.P, D=p.D, data_bins=p.data_carriers, const_points=p.const_points, const_bits=p.const_bits,
                   known_bits=p.known_bits, in_dtype=dt, fit_lo=p.fit_lo, fit_hi=p.fit_hi, max_window=min(wmax, N // 2), thresh=thresh)
    if W > min(wmax, N // 2): W = min(wmax, N // 2)
    eng = Engine(cfg)
    payload = rs.randint(0, 2, F * p.D * p.C * p.mu)
    fill = rs.choice(np.array([1 + 1j, 1 - 1j, -1 + 1j, -1 - 1j]) / np.sqrt(2), size=p.K - p.C)
    filler = np.zeros(p.K, dtype=complex)
    filler[np.delete(np.arange(1, p.K + 1), p.data_carriers - 1) - 1] = fill
    gaps = rs.randint(0, max(1, W - 2), F)
    stride = p.frame_len + wmax + 64
    rows = eng.tx_frames(orc.pack_bits(payload, p.D * p.C * p.mu), filler, stride=stride, gaps=gaps, out_dtype=torch.float64)
    rows = rows + torch.from_numpy(rs.randn(*rows.shape)).cuda() * float(rows.std()) * 10 ** (-snr / 20)
    kind = rs.randint(0, 5)
    tt = torch.arange(rows.shape[1], device="cuda", dtype=torch.float64)
    if kind == 1: rows = rows + 0.5 * torch.sin(2 * np.pi * rs.uniform(0.3, 0.49) * tt)
    if kind == 2: rows = rows + 0.1 * torch.sin(2 * np.pi * rs.uniform(0.001, 0.15) * tt)
    if kind == 3: rows = -rows
    if kind == 4 and F > 1: rows[rs.randint(0, F)] = 0.0
    if storage == "int16": rows = torch.round(rows * (20000.0 / float(rows.abs().max()))).to(torch.int16)
    elif storage == "uint8": rows = (torch.round(rows * (100.0 / float(rows.abs().max()))) + 128).to(torch.uint8)
    else: rows = rows.to(dt)
    lo = int(rs.randint(-10, 3))
    a = eng.sync_frames(rows, F, stride, lo, lo + W)
    work = eng.sync_frames_workspace(F)
    b = eng.sync_frames(rows, F, stride, lo, lo + W, screened=True, work=work)
    nun = int(work[:4].view(torch.int32).item())
    ok = bool(torch.equal(a, b))
    if case % 5 == 0:                                           # the bound itself, per lag, on this case's windows
        d = eng.debug_frames_screen(rows, F, stride, lo, lo + W)
        r64 = rows.cpu().numpy().astype(np.float64).reshape(-1)
        c = orc.chirp_replica(p)
        for f in range(F):
            s0 = f * stride + lo
            seg = np.zeros(W + p.Lc - 1); l0, h0 = max(0, s0), min(len(r64), s0 + len(seg))
            if h0 > l0: seg[l0 - s0: h0 - s0] = r64[l0:h0]
            y = np.correlate(seg, c, mode="valid")
            e = float(np.abs(d["y32"][f].cpu().numpy().astype(np.float64) - y).max()); eb = float(d["err"][f])
            worst = max(worst, e / eb if eb > 0 else 0.0)
            ok = ok and e <= eb
    windows += F; unresolved += nun
    if not ok:
        bad += 1
        print("MISMATCH", case, dict(N=N, F=F, cp=cp, storage=storage, snr=snr, thresh=thresh, W=W, lo=lo, kind=int(kind)), a.tolist(), b.tolist(), nun, flush=True)
    eng.close()
    if case % 20 == 19: print("case", case + 1, "elapsed", round(time.time() - t0, 1), "mismatches", bad, "windows", windows, "to fp64", unresolved, "worst realised/bound", round(worst, 3), flush=True)
print("cases", ncase, "mismatches", bad, "windows", windows, "sent to fp64", unresolved, "worst realised/bound", round(worst, 3))
