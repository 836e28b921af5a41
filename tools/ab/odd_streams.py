#!/usr/bin/env python3
"""Odd streams through every sync path against the oracle: negative constants, inverted chirps, NaN / Inf samples."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import gf3_oracle as orc
from tests.test_gpu_parity import load, params_of, engine_for
g = load("g1_n1024_qpsk"); p = params_of(g); c = orc.chirp_replica(p); rs = np.random.RandomState(4)
n = 60000
base = 0.01 * rs.randn(n); base[5000:5000 + p.Lc] += c; base[30000:30000 + p.Lc] += c
cases = {"neg_const": np.full(40000, -3.0), "pos_const": np.full(40000, 3.0), "inverted": -base, "neg_dc": base - 5.0,
         "nan": base.copy(), "inf": base.copy(), "ninf": base.copy(), "neg_only_peak": -np.abs(base)}
cases["nan"][12345] = np.nan; cases["inf"][23456] = np.inf; cases["ninf"][23456] = -np.inf
eng = engine_for(p)
bad = 0
for name, r in cases.items():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = np.flatnonzero(orc.chirp_method(r, p))
    for mode in (1, 2, 3):
        eng.sync_stream_mode(mode)
        try:
            got = eng.sync_stream(torch.from_numpy(r).cuda(), cap=len(r) + p.Lc).cpu().numpy()
            ok = np.array_equal(got, want)
        except Exception as e:
            got, ok = str(e)[:80], False
        bad += not ok
        print(name, "mode", mode, "ok" if ok else "MISMATCH", len(want), eng.sync_stream_info(), "" if ok else (got[:8] if hasattr(got, "__len__") else got, want[:8]), flush=True)
print("mismatches:", bad)
