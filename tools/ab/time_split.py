#!/usr/bin/env python3
"""One launch against the two-phase form of gf3_demod_frames_ex over F and the packet geometry (HIP events, median of 10 after 2):
where the library's dispatch rule should put the line.  Geometries: the reference's (mode A2: P = 20, D = 180, N = 4096), a
medium one (P = 4, D = 40) and the bench's (P = 2, D = 8)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table
known = np.unpackbits(np.load(os.path.join(ROOT, "gf3_audio_modem_amd", "data", "known_bits.npz"))["packed"])

def ev(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in es:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in es]))

pts, bt = qpsk_table()
for name, N, CP, P, D, lo, hi in (("A2", 4096, 224, 20, 180, 100, 1500), ("mid", 4096, 512, 4, 40, 1, 2047), ("bench", 4096, 512, 2, 8, 1, 2047), ("N1024", 1024, 128, 4, 60, 1, 511)):
    K = N // 2 - 1
    cfg = RxConfig(N=N, CP=CP, P=P, D=D, data_bins=np.arange(lo, hi), const_points=pts, const_bits=bt, known_bits=np.tile(known, 2), in_dtype=torch.float32)
    eng = Engine(cfg)
    for F in (1, 3, 8, 16, 32, 64, 128, 256, 512, 1024):
        if F * cfg.frame_len * 4 > 8e9: break
        gen = torch.Generator(device="cuda").manual_seed(F)
        packed = torch.randint(0, 256, (F, eng.bytes_per_frame), dtype=torch.uint8, device="cuda", generator=gen)
        filler = np.zeros(K, dtype=complex); filler[np.delete(np.arange(K), np.arange(lo, hi) - 1)] = (1 - 1j) / np.sqrt(2)
        rows = eng.tx_frames(packed, filler, out_dtype=torch.float32)
        starts = torch.arange(F, device="cuda") * rows.shape[1] + cfg.chirp_length
        o1 = eng.demod_frames(rows, starts, split=False)["bits"]; o2 = eng.demod_frames(rows, starts, split=True)["bits"]
        t1 = ev(lambda: eng.demod_frames(rows, starts, split=False)); t2 = ev(lambda: eng.demod_frames(rows, starts, split=True))
        ta = ev(lambda: eng.demod_frames(rows, starts))
        plan = eng.demod_plan(F, split=True); auto = eng.demod_plan(F)
        print(json.dumps(dict(geom=name, F=F, one_launch_ms=round(t1, 4), two_phase_ms=round(t2, 4), auto_ms=round(ta, 4), Dc=plan["Dc"], chunks=plan["chunks"], auto_split=auto["split"],
                              same=bool(torch.equal(o1, o2)), payload=bool(torch.equal(o1, packed)))), flush=True)
        del rows, packed
    eng.close()
