#!/usr/bin/env python3
"""Random geometries / storage / noise / thresholds through every stream-sync path -- the screened call, the general
screening kernel, the all-fp64 call, and the stream taken from host memory in pieces -- against the oracle (the property
test's generator with other seeds, more cases, lower thresholds and occasional interferers).  argv[1] = cases, argv[2] = seed."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, dataclasses
from oracle import gf3_oracle as orc
from tests.test_properties import _params
from gf3_audio_modem_amd import Engine, RxConfig
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
bad = 0
t0 = time.time()
for case in range(ncase):
    N = int(rs.choice([1024, 2048, 4096, 8192])); F = int(rs.randint(1, 4)); cp = float(rs.choice([1 / 32, 1 / 8, 1 / 4, 1 / 2]))
    storage = str(rs.choice(["float64", "float32", "int16", "uint8"])); snr = float(rs.choice([60.0, 20.0, 6.0, 0.0]))
    thresh = float(rs.choice([0.4, 0.4, 0.25, 0.6, 0.9]))
    p = dataclasses.replace(_params(N, cp, 1, 2, 2, 0.0, 0.0), thresh=thresh)
    payload = rs.randint(0, 2, F * p.D * p.C * p.mu)
    fill = rs.choice(np.array([1 + 1j, 1 - 1j, -1 + 1j, -1 - 1j]) / np.sqrt(2), size=p.K - p.C)
    r = orc.tx_stream(payload, fill, p, gaps=rs.randint(0, 400, F), lead=int(rs.randint(0, 3000)), tail=int(rs.randint(0, 500)))
    r = r + rs.randn(len(r)) * np.sqrt(np.mean(r * r)) * 10 ** (-snr / 20)
    kind = rs.randint(0, 5)
    if kind == 1: r = r + 0.5 * np.sin(2 * np.pi * rs.uniform(0.3, 0.49) * np.arange(len(r)))
    if kind == 2: r = r + 0.1 * np.sin(2 * np.pi * rs.uniform(0.001, 0.15) * np.arange(len(r)))
    if kind == 3: r = -r
    if storage == "int16": rq = np.round(r / np.abs(r).max() * 20000).astype(np.int16)
    elif storage == "uint8": rq = np.round(r / np.abs(r).max() * 100 + 128).astype(np.uint8)
    else: rq = r.astype(storage)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = np.flatnonzero(orc.chirp_method(rq.astype(np.float64), p))
    cfg = RxConfig(N=p.N, CP=p.CP, P=p.P, D=p.D, data_bins=p.data_carriers, const_points=p.const_points, const_bits=p.const_bits,
                   known_bits=p.known_bits, in_dtype=getattr(torch, storage), fit_lo=p.fit_lo, fit_hi=p.fit_hi, thresh=thresh)
    eng = Engine(cfg)
    x = torch.from_numpy(rq).cuda()
    for mode in (2, 3, 1):
        eng.sync_stream_mode(mode)
        got = eng.sync_stream(x, cap=len(rq) + p.Lc).cpu().numpy()
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH", case, dict(N=N, F=F, cp=cp, storage=storage, snr=snr, thresh=thresh, kind=int(kind), mode=mode), eng.sync_stream_info(), got[:6], want[:6], flush=True)
    # ... and the stream taken from host memory in pieces of a random size (Engine.receive_host): same detections, or the
    # reference's own failure (fewer than two detections; a non-last packet that runs past the end)
    L = p.M * p.S
    fails = len(want) < 2 or bool(np.any(want[:-1] + 2 + L > len(rq)))
    try:
        out = eng.receive_host(rq, chunk_samples=int(rs.uniform(2.0, 5.0) * p.frame_len))
        ok = (not fails) and np.array_equal(out["peaks"].cpu().numpy(), want)
        info = out["info"]
    except ValueError as e:
        ok, info = fails, str(e)
    if not ok:
        bad += 1
        print("MISMATCH (host ingest)", case, dict(N=N, F=F, cp=cp, storage=storage, snr=snr, thresh=thresh, kind=int(kind)), info, want[:6], flush=True)
    eng.close()
    if case % 10 == 9: print("case", case + 1, "elapsed", round(time.time() - t0, 1), "mismatches", bad, flush=True)
print("cases", ncase, "mismatches", bad)
