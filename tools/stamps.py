#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the fused demod kernel (needs a library built with
-DGF3_STAMPS: GF3_LIB=/path/to/lib python tools/stamps.py).  Never quote this build's run time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table

N, CP, P, D, F = 4096, 512, 2, 8, 16384
K = N // 2 - 1
pts, bt = qpsk_table()
known = np.unpackbits(np.load(os.path.join(os.path.dirname(__file__), "..", "gf3_audio_modem_amd", "data", "known_bits.npz"))["packed"])
cfg = RxConfig(N=N, CP=CP, P=P, D=D, data_bins=np.arange(1, K), const_points=pts, const_bits=bt, known_bits=known,
               in_dtype=torch.float32, max_window=320)
eng = Engine(cfg)
stride = 78720
gen = torch.Generator(device="cuda").manual_seed(1)
payload = torch.randint(0, 256, (F, eng.bytes_per_frame), dtype=torch.uint8, device="cuda", generator=gen)
gaps = torch.randint(0, 300, (F,), dtype=torch.int64, device="cuda", generator=gen)
filler = np.zeros(K, dtype=complex); filler[K - 1] = pts[0]
big = eng.tx_frames(payload, filler, stride=stride, gaps=gaps, out_dtype=torch.float32)
starts = eng.sync_frames(big, F, stride, -8, 312)
st = torch.zeros((F, 8), dtype=torch.int64, device="cuda")
eng.lib.gf3_debug_set_stamps(eng._h, C.c_void_p(st.data_ptr()))
for _ in range(3):
    eng.demod_frames(big, starts)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
d = np.diff(s[:, :6], axis=1)
names = ["start + start-pilot sum", "start + end pilot transforms", "finalize (H, angles, slope)", "data symbols", "last pack + exit"]
tot = s[:, 5] - s[:, 0]
print("median block lifetime (s_memtime ticks):", np.median(tot))
for i, nme in enumerate(names):
    print(f"  {nme:32s} median {np.median(d[:, i]):10.0f}  share {np.median(d[:, i]) / np.median(tot):6.1%}")

# finer stamps inside finalize when present (slots 6, 7 of a build that sets them)
if s[:, 6].max() > 0:
    print("  finalize: per-carrier H/angles   median", np.median(s[:, 6] - s[:, 2]))
    print("  finalize: barrier + slope sums   median", np.median(s[:, 7] - s[:, 6]))
    print("  finalize: rotation-table init    median", np.median(s[:, 3] - s[:, 7]))
