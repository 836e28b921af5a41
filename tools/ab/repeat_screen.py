#!/usr/bin/env python3
"""Run the screening pass repeatedly on the config-3 stream and compare the block bounds / maxima between runs."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("c3", os.path.join(ROOT, "tools", "config3.py"))
c3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c3)
eng, cfg, channel = c3.make_engine()
r, payload = c3.make_stream(eng, channel, 4096)
eng.sync_stream_mode(2)
ref = None
for i in range(8):
    p32, bmax, berr, hop = eng.debug_stream_screen(r)
    be = berr.cpu().numpy(); bm = bmax.cpu().numpy()
    nf = np.flatnonzero(~np.isfinite(be))
    print(i, "non-finite bounds:", len(nf), nf[:6], "max", float(np.nanmax(be)), flush=True)
    if ref is None: ref = (be, bm)
    else:
        d = np.flatnonzero((be != ref[0]) | (bm != ref[1]))
        print("   blocks differing from run 0:", len(d), d[:8], [(float(ref[0][k]), float(be[k])) for k in d[:3]], flush=True)
    pk = eng.sync_stream(r); print("   sync path", eng.sync_stream_info(), flush=True)
