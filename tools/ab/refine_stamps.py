#!/usr/bin/env python3
"""Diagnostic: where a wave of scr_refine_kernel spends its time on the config-3 stream (needs a -DGF3_STAMPS build:
GF3_LIB=/path/lib.so python tools/ab/refine_stamps.py).  The stamps serialise the phases of a step (each waits for all
LDS traffic of the wave), so the sum is an upper bound of the production kernel's step time; never quote it as a run time."""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
spec = importlib.util.spec_from_file_location("gf3_config3", os.path.join(ROOT, "tools", "config3.py"))
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
eng, cfg, channel = tool.make_engine()
r, payload = tool.make_stream(eng, channel, 4096)
nw = 2 * 256 * 4
st = torch.zeros((nw, 8), dtype=torch.int64, device="cuda")
for _ in range(3):
    eng.sync_stream(r, 8192)
eng.lib.gf3_debug_set_stamps(eng._h, C.c_void_p(st.data_ptr()))
eng.sync_stream(r, 8192)
torch.cuda.synchronize()
eng.lib.gf3_debug_set_stamps(eng._h, C.c_void_p(0))
s = st.cpu().numpy().astype(np.float64)
s = s[s[:, 7] > 0]
life = s[:, 0]
print("waves that worked: %d; steps per wave: median %.0f (min %.0f max %.0f)" % (len(s), np.median(s[:, 7]), s[:, 7].min(), s[:, 7].max()))
print("wave lifetime: median %.0f ticks, max %.0f; per step %.0f" % (np.median(life), life.max(), np.median(life / s[:, 7])))
wall = (s[:, 2] - s[:, 1]) * 10.0            # ns (s_memrealtime runs at 100 MHz)
print("wave wall time: median %.1f us, max %.1f us; first start -> last end %.1f us; shader clock = ticks / wall: median %.0f MHz" % (
    np.median(wall) / 1e3, wall.max() / 1e3, (s[:, 2].max() - s[:, 1].min()) * 10.0 / 1e3, np.median(life / wall) * 1e3))
names = ["step body (reads, stores, loads, conversions, 256 fma)", "cell tail (reduce, store), per step", "next item (cell numbers, draw)"]
for i, nme in enumerate(names):
    per = s[:, 3 + i] / s[:, 7]
    print("  %-56s per step: median %7.0f ticks  p90 %7.0f   share of lifetime %5.1f %%" % (nme, np.median(per), np.percentile(per, 90), 100 * np.median(s[:, 3 + i] / life)))
