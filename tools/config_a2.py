#!/usr/bin/env python3
"""The reference's own default geometry (mode A2: N=4096, CP=224, P=20, D=180, data bins 100..1499, QPSK) at batch
scale: F packets synthesised on the device, windowed chirp sync + fused demodulation, every bit checked.
Reports samples/s (not the BASELINE metric's configuration; a data point for DESIGN.md)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=2048)
args = ap.parse_args()
F = args.frames
N, CP, P, D = 4096, 224, 20, 180
K = N // 2 - 1
pts, bt = qpsk_table()
known = np.unpackbits(np.load(os.path.join(ROOT, "gf3_audio_modem_amd", "data", "known_bits.npz"))["packed"])
known = np.tile(known, -(-K * 2 // len(known)))
bins = np.arange(100, 1500)                                         # mode "A2": lowest_bin 100, highest_bin 1500 exclusive (OFDM.py:32,47)
cfg = RxConfig(N=N, CP=CP, P=P, D=D, data_bins=bins, const_points=pts, const_bits=bt, known_bits=known,
               in_dtype=torch.float32, max_window=320)
eng = Engine(cfg)
gen = torch.Generator(device="cuda").manual_seed(7)
payload = torch.randint(0, 256, (F, eng.bytes_per_frame), dtype=torch.uint8, device="cuda", generator=gen)
gaps = torch.randint(0, 300, (F,), dtype=torch.int64, device="cuda", generator=gen)
rs = np.random.RandomState(1)
filler = np.zeros(K, dtype=complex)
unused = np.delete(np.arange(1, K + 1), bins - 1)
filler[unused - 1] = rs.choice(pts, size=len(unused))
stride = cfg.frame_len + 320
rows = eng.tx_frames(payload, filler, stride=stride, gaps=gaps, out_dtype=torch.float32)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
best = None
for _ in range(4):
    ev[0].record(); starts = eng.sync_frames(rows, F, stride, -8, 312); ev[1].record()
    bits = eng.demod_frames(rows, starts)["bits"]; ev[2].record(); torch.cuda.synchronize()
    t = (ev[0].elapsed_time(ev[1]) * 1e-3, ev[1].elapsed_time(ev[2]) * 1e-3)
    best = t if best is None or sum(t) < sum(best) else best
ok_sync = bool(torch.equal(starts, torch.arange(F, device="cuda") * stride + gaps + cfg.chirp_length))
errs = int(torch.count_nonzero(torch.bitwise_xor(bits, payload)).item())
n = F * stride
print(json.dumps({"config": "reference default geometry (mode A2, P=20, D=180)", "frames": F, "samples": n, "sync_s": best[0], "demod_s": best[1],
                  "samples_per_s": n / sum(best), "differing_bytes": errs, "sync_exact": ok_sync,
                  "demod_algorithmic_GBps": F * (4 * cfg.M * N + eng.bytes_per_frame) / best[1] / 1e9}))
