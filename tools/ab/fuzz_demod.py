#!/usr/bin/env python3
"""Random geometries / constellations / storage / noise through the fused demodulation kernel (sync_frames + demod_frames, all
three modes: the QPSK sign rule, the bits-only table mode and the table mode with dumps) against the oracle on the same
samples: bits identical, equalised symbols within 1e-9 -- and, since round 4, every case again through the TWO-PHASE form
(gf3_demod_frames_ex, split=True: pilot sums, estimate, data symbols in chunks; D up to 40 so that packets cut into several
chunks): its bits, Hs-derived slope and equalised symbols against the one-launch kernel's.  argv[1] = cases, argv[2] = seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import gf3_oracle as orc
from tests.test_properties import _params
from gf3_audio_modem_amd import Engine, RxConfig
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad, t0 = 0, time.time()
for case in range(ncase):
    N = int(rs.choice([1024, 2048, 4096, 8192])); F = int(rs.randint(1, 4)); cp = float(rs.choice([1 / 32, 1 / 8, 1 / 4]))
    P, D, mu = int(rs.randint(1, 4)), int(rs.choice([1, 2, 3, 4, 4, 9, 16, 23, 40])), int(rs.choice([2, 4, 4, 6, 6]))
    storage = str(rs.choice(["float64", "float32", "int16"])); snr = float(rs.choice([60.0, 30.0, 20.0, 12.0]))
    p = _params(N, cp, P, D, mu, float(rs.uniform()), float(rs.uniform()))
    dt = getattr(torch, storage)
    cfg = RxConfig(N=p.N, CP=p.CP, P=p.P, D=p.D, data_bins=p.data_carriers, const_points=p.const_points, const_bits=p.const_bits,
                   known_bits=p.known_bits, in_dtype=dt, fit_lo=p.fit_lo, fit_hi=p.fit_hi, max_window=256)
    eng = Engine(cfg)
    payload = rs.randint(0, 2, F * p.D * p.C * p.mu)
    packed = orc.pack_bits(payload, p.D * p.C * p.mu)
    fill = rs.choice(np.array([1 + 1j, 1 - 1j, -1 + 1j, -1 - 1j]) / np.sqrt(2), size=p.K - p.C)
    filler = np.zeros(p.K, dtype=complex)
    filler[np.delete(np.arange(1, p.K + 1), p.data_carriers - 1) - 1] = fill
    gaps = rs.randint(0, 200, F)
    stride = p.frame_len + 256
    rows = eng.tx_frames(packed, filler, stride=stride, gaps=gaps, out_dtype=torch.float64)
    rows = rows + torch.from_numpy(rs.randn(*rows.shape)).cuda() * float(rows.std()) * 10 ** (-snr / 20)
    # a slowly varying channel gain per row and a small clock drift keep |He| != |Hs| and the slope non-zero
    rows = rows * torch.from_numpy(1.0 + 0.2 * rs.rand(F, 1)).cuda()
    if storage == "int16":
        rows = torch.round(rows * (20000.0 / float(rows.abs().max()))).to(torch.int16)
    else:
        rows = rows.to(dt)
    starts = eng.sync_frames(rows, F, stride, -8, 248)
    x = rows.cpu().numpy().astype(np.float64).reshape(-1)
    st = starts.cpu().numpy()
    if (st < 0).any():
        eng.close(); continue                                   # (sync found nothing in the window at this SNR: not this tool's subject)
    ref = orc.demod_frames(x, st, p)
    lean = eng.unpack_bits(eng.demod_frames(rows, starts)["bits"]).cpu().numpy()
    full = eng.demod_frames(rows, starts, want=("eq", "slope"))
    fb = eng.unpack_bits(full["bits"]).cpu().numpy()
    err = float(np.abs(full["eq"].cpu().numpy() - ref["eq"]).max() / max(1.0, np.abs(ref["eq"]).max()))
    lean2 = eng.demod_frames(rows, starts, split=True)["bits"]
    full2 = eng.demod_frames(rows, starts, want=("eq", "slope"), split=True)
    err2 = float((full2["eq"] - full["eq"]).abs().max() / max(1.0, float(full["eq"].abs().max())))
    same = (np.array_equal(eng.unpack_bits(lean2).cpu().numpy(), lean) and torch.equal(full2["bits"], full["bits"])
            and torch.equal(full2["slope"], full["slope"]) and err2 < 1e-12)
    ok = np.array_equal(lean, ref["bits"].reshape(-1)) and np.array_equal(fb, ref["bits"].reshape(-1)) and err < 1e-9 and same
    if not ok:
        bad += 1
        nd = int(np.sum(lean != ref["bits"].reshape(-1)))
        print("MISMATCH", case, dict(N=N, F=F, cp=cp, P=P, D=D, mu=mu, storage=storage, snr=snr), "bits differing (lean)", nd,
              "(full)", int(np.sum(fb != ref["bits"].reshape(-1))), "eq err", err, "two-phase == one-launch:", same, "eq diff", err2, eng.demod_plan(F, split=True), flush=True)
    eng.close()
    if case % 10 == 9: print("case", case + 1, "elapsed", round(time.time() - t0, 1), "mismatches", bad, flush=True)
print("cases", ncase, "mismatches", bad)
