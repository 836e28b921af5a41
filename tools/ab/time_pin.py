#!/usr/bin/env python3
"""How to get a PAGEABLE host array to the device fastest: staged copies (1 thread / N threads) vs pinning it in place."""
import time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 320887424
a = np.random.default_rng(0).standard_normal(n, dtype=np.float32)
t_a = torch.from_numpy(a)
stage = torch.empty(1 << 25, dtype=torch.float32).pin_memory()
dst = torch.empty(n, dtype=torch.float32, device="cuda")
def staged(threads):
    pool = ThreadPoolExecutor(threads) if threads > 1 else None
    torch.cuda.synchronize(); t = time.perf_counter()
    for lo in range(0, n, 1 << 25):
        hi = min(n, lo + (1 << 25)); m = hi - lo
        if pool:
            step = -(-m // threads)
            list(pool.map(lambda k: stage[k * step: min(m, (k + 1) * step)].copy_(t_a[lo + k * step: min(hi, lo + (k + 1) * step)]), range(threads)))
        else:
            stage[:m].copy_(t_a[lo:hi])
        dst[lo:hi].copy_(stage[:m], non_blocking=True); torch.cuda.synchronize()
    return time.perf_counter() - t
for th in (1, 4, 8, 16):
    staged(th); dt = staged(th)
    print("staged, %2d host thread(s): %.1f ms = %.1f GB/s" % (th, dt * 1e3, n * 4 / dt / 1e9), flush=True)
for piece in (1 << 23, 1 << 25, 1 << 27):                       # the runtime's own path for pageable memory (it stages or pins by itself)
    torch.cuda.synchronize(); t = time.perf_counter()
    for lo in range(0, n, piece):
        dst[lo:lo + piece].copy_(t_a[lo:lo + piece])
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("plain copy_ from the pageable array in pieces of %9d samples: %.1f ms = %.1f GB/s" % (piece, dt * 1e3, n * 4 / dt / 1e9), flush=True)
rt = torch.cuda.cudart()
t = time.perf_counter(); rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t_reg = time.perf_counter() - t
print("cudaHostRegister rc", rc, "%.1f ms" % (t_reg * 1e3), flush=True)
torch.cuda.synchronize(); t = time.perf_counter(); dst.copy_(t_a, non_blocking=True); torch.cuda.synchronize(); t_cp = time.perf_counter() - t
print("copy from the registered array: %.1f ms = %.1f GB/s (is_pinned %s)" % (t_cp * 1e3, n * 4 / t_cp / 1e9, t_a.is_pinned()), flush=True)
t = time.perf_counter(); rt.cudaHostUnregister(a.ctypes.data); print("unregister %.1f ms" % ((time.perf_counter() - t) * 1e3))
