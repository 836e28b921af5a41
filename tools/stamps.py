#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the fused demod kernel (needs a library built with
-DGF3_STAMPS: GF3_LIB=/path/to/lib python tools/stamps.py).  Never quote this build's run time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table

N, CP, P, D, F = 4096, 512, 2, 8, int(os.environ.get("GF3_STAMPS_FRAMES", "16384"))
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
import time, importlib.util
_spec = importlib.util.spec_from_file_location("gf3_bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
_bench = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_bench)
_ps = _bench.PowerSampler(0); _ps.__enter__()
t0 = time.perf_counter()
while time.perf_counter() - t0 < float(os.environ.get("GF3_STAMPS_SECONDS", "2.5")):     # sustained load: the clock settles after ~2 s
    for _ in range(20):
        eng.demod_frames(big, starts)
    torch.cuda.synchronize()
_ps.__exit__()
if _ps.samples:
    print(f"package power while looping: median of the second half {np.median(_ps.samples[len(_ps.samples) // 2:]):.0f} W over {len(_ps.samples)} samples")
s = st.cpu().numpy().astype(np.float64)
d = np.diff(s[:, :6], axis=1)
names = ["start + start-pilot sum", "start + end pilot transforms", "finalize (H, angles, slope)", "data symbols", "last pack + exit"]
tot = s[:, 5] - s[:, 0]
print("median block lifetime (s_memtime ticks):", np.median(tot))
for i, nme in enumerate(names):
    print(f"  {nme:32s} median {np.median(d[:, i]):10.0f}  share {np.median(d[:, i]) / np.median(tot):6.1%}")

# slots 6, 7: s_memrealtime (100 MHz) at the first and last stamp of the packet -> shader clock and wall time
if s[:, 6].max() > 0:
    wall = (s[:, 7] - s[:, 6]) * 10.0                       # ns
    print(f"packet wall time median {np.median(wall) / 1e3:.2f} us; shader clock {np.median(tot / wall) * 1e3:.0f} MHz")
    print(f"first stamp -> last stamp over the whole launch: {(s[:, 7].max() - s[:, 6].min()) * 10.0 / 1e6:.3f} ms for {F} packets")

# persistent kernels: packets f, f + G, f + 2G ... belong to one workgroup (G = resident grid)
G = int(os.environ.get("GF3_STAMPS_GRID", "0"))
if G:
    r0 = s[:, 6].reshape(-1, G) * 10.0      # [round, workgroup] ns
    r1 = s[:, 7].reshape(-1, G) * 10.0
    gap = r0[1:] - r1[:-1]
    print(f"persistent grid {G}: gap between packets of one workgroup: median {np.median(gap) / 1e3:.2f} us, p90 {np.percentile(gap, 90) / 1e3:.2f} us")
    t_end = r1[-1] - r0[0].min()
    t_beg = r0[0] - r0[0].min()
    print(f"  workgroup start skew: median {np.median(t_beg) / 1e3:.1f} us, max {t_beg.max() / 1e3:.1f} us; finish: min {t_end.min() / 1e3:.1f} median {np.median(t_end) / 1e3:.1f} max {t_end.max() / 1e3:.1f} us")
    per = (r1 - r0)
    print(f"  packet wall time by round (median over workgroups, us):", np.round(np.median(per, axis=1) / 1e3, 1)[:40])
    fin = r1[-1] - r0[0].min()
    print("  finish time (us) by XCD (blockIdx % 8):", np.round([np.median(fin[x::8]) / 1e3 for x in range(8)], 0))
    print("  finish time (us) by grid half:", np.round([np.median(fin[:G // 2]) / 1e3, np.median(fin[G // 2:]) / 1e3], 0))
    within = fin.reshape(-1, 8)            # [slot in XCD, XCD]
    print("  finish time (us) by position inside XCD 0 (blockIdx // 8):", np.round(within[:, 0] / 1e3, 0))
