#!/usr/bin/env python3
"""Developer check of the band-limited screening kernel (scr_ring_kernel) against the general one (mode 3) and the
oracle, N=4096 geometries only (works with GF3_DEV_BUILD libraries; GF3_LIB selects the build)."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import gf3_oracle as orc
spec = importlib.util.spec_from_file_location("c3", os.path.join(ROOT, "tools", "config3.py"))
c3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c3)
eng, cfg, channel = c3.make_engine()

def ev_ms(f, reps=5):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return out, ts

# 1. bound against the oracle on a short stream (8 packets through the channel)
r8, _ = c3.make_stream(eng, channel, 8)
from tests.test_gpu_parity import params_of  # noqa: E402  (parameter block of a fixture)
g = np.load(os.path.join(ROOT, "tests", "golden", "g3_n4096_16qam_gr5.npz"))
p = params_of(g)
P = orc.matched_filter(r8.cpu().numpy().astype(np.float64), p)
cases = [("c3x8", r8)]
rs = np.random.RandomState(3)
spike = np.zeros(200000, dtype=np.float32); spike[91234] = 1e6; spike[5] = -3e5
cases.append(("spike", torch.from_numpy(spike).cuda()))
cases.append(("noise", torch.from_numpy(rs.randn(300000).astype(np.float32)).cuda()))
cases.append(("ones", torch.ones(100000, dtype=torch.float32, device="cuda")))
for name, x in cases:
    Pn = orc.matched_filter(x.cpu().numpy().astype(np.float64), p)
    for mode in (2, 3):
        eng.sync_stream_mode(mode)
        p32, bmax, berr, hop = eng.debug_stream_screen(x)
        p32 = p32.cpu().numpy().astype(np.float64); berr = berr.cpu().numpy().astype(np.float64)
        tol = 1e-13 * np.linalg.norm(x.cpu().numpy().astype(np.float64)) * np.linalg.norm(orc.chirp_replica(p))
        err = np.maximum(np.abs(p32 - Pn) - tol, 0)
        per = np.repeat(berr, hop)[: len(Pn)]
        want_max = np.array([p32[b * hop:(b + 1) * hop].max() for b in range(len(berr))])
        print(name, "mode", mode, "hop", hop, "max realised/bound", float((err / per).max()), "bound/max|P|", float(berr.max() / np.abs(Pn).max()),
              "bmax ok", bool(np.array_equal(bmax.cpu().numpy().astype(np.float64), want_max)), flush=True)

# 2. the config-3 stream: both kernels agree within the sum of their bounds on every lag; timings; peaks
r, payload = c3.make_stream(eng, channel, 4096)
res = {}
for mode in (2, 3):
    eng.sync_stream_mode(mode)
    (p32, bmax, berr, hop), ts = ev_ms(lambda: eng.debug_stream_screen(r))
    res[mode] = (p32, bmax, berr, hop, ts)
    print("mode", mode, "screen ms", [round(t, 3) for t in ts], flush=True)
(pa, ma, ea, hop, _), (pb, mb, eb, _, _) = res[2], res[3]
d = (pa - pb).abs()
lim = (ea + eb).repeat_interleave(hop)[: d.numel()]
print("ring vs general: max |diff| / (E_ring + E_gen) =", float((d / lim).max()), " max E_ring/E_gen", float((ea / eb).max()),
      " max E_ring", float(ea.max()), "max P32", float(pa.max()), flush=True)
peaks = {}
for mode in (2, 3, 1):
    eng.sync_stream_mode(mode)
    pk, ts = ev_ms(lambda: eng.sync_stream(r), reps=4)
    peaks[mode] = pk.cpu().numpy()
    print("mode", mode, "sync_stream ms", [round(t, 3) for t in ts], eng.sync_stream_info(), flush=True)
print("peaks equal:", bool(np.array_equal(peaks[2], peaks[1]) and np.array_equal(peaks[3], peaks[1])), len(peaks[1]))
