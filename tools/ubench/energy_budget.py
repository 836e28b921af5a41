#!/usr/bin/env python3
"""Driver of tools/ubench/energy_budget.hip (diagnostic, run on the GPU box): package power (amdgpu hwmon, sampled as
bench.py samples it) of instruction streams at several issue densities and of streaming reads from HBM / Infinity
Cache / L2, and from those the line  P = P_awake + rate * e_op  per instruction class.  Prints one JSON document;
DESIGN.md section 8.0 has the budget built from it.

    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libenergy.so tools/ubench/energy_budget.hip
    python tools/ubench/energy_budget.py > gpurun_out/energy_budget.json
"""
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("gf3_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

lib = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libenergy.so"))
lib.eb_spin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double)]
lib.eb_stream.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double)]
SEC = float(os.environ.get("EB_SECONDS", "3.0"))
OPS = {"fma64": 0, "add64": 1, "addu32": 2, "ldsr": 3, "ldsw": 4, "sleep": 5, "pad_only": 6, "mul64": 7, "cndmask": 8, "mov": 9,
       "salu": 10, "cvt": 11, "fma64_random": 12, "add64_random": 13, "mul64_random": 14, "ldsr_random": 15, "ldsw_random": 16}


def powered(fn):
    ps = bench.PowerSampler(0)
    with ps:
        r = fn()
    s = ps.samples
    return r, (float(np.median(s[len(s) // 2:])) if s else None)


def spin(op, blocks, lds, pad, iters=100000):
    out = (C.c_double * 3)()
    rc, w = powered(lambda: lib.eb_spin(OPS[op], blocks, lds, iters, pad, SEC, out))
    assert rc == 0, (op, rc)
    return {"op": op, "workgroups": blocks, "lds_bytes": lds, "pad_s_nop15": pad, "lane_ops_per_s": out[1], "clock_MHz": out[2], "package_w": w}


def stream(name, buf, per_launch, wrap, blocks=2048, random_data=0):
    out = (C.c_double * 2)()
    rc, w = powered(lambda: lib.eb_stream(buf, per_launch, wrap, blocks, random_data, SEC, out))
    assert rc == 0, (name, rc)
    return {"stream": name, "buffer_bytes": buf, "wrap_bytes_per_xcd": wrap, "random_data": bool(random_data), "bytes_per_s": out[1], "package_w": w}


def main():
    if os.environ.get("EB_STREAMS_ONLY"):
        torch.cuda.init()
        G = 1 << 30
        _, idle = powered(lambda: time.sleep(SEC))
        out = [spin("sleep", 512, 40960, 0, iters=2000)] + [stream(n, 16 * G, 16 * G, 0, random_data=r) for n, r in
               (("hbm", 0), ("hbm_random", 1), ("hbm_8byte_loads", 2), ("hbm_random_8byte_loads", 3))]
        print(json.dumps({"idle_w": idle, "points": out}, indent=1))
        return
    torch.cuda.init()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    res = {"n_cu": n_cu, "seconds_per_point": SEC, "points": []}
    _, idle = powered(lambda: time.sleep(SEC))
    res["idle_w"] = idle
    one, two = 98304, 40960                     # dynamic LDS per workgroup: one / two workgroups of 4 waves per CU
    P = res["points"]
    P.append(spin("sleep", 2 * n_cu, two, 0, iters=2000))
    P.append(spin("pad_only", 2 * n_cu, two, 4))
    for pad in (0, 1, 2, 4, 8):
        P.append(spin("fma64", n_cu, one, pad))
    for pad in (0, 2, 4, 8, 16):
        P.append(spin("fma64", 2 * n_cu, two, pad))
    for op in ("fma64_random", "add64_random", "mul64_random"):
        for pad in (0, 2, 8):
            P.append(spin(op, 2 * n_cu, two, pad, iters=50000))
    for op in ("add64", "addu32", "mul64", "cndmask", "mov", "salu", "cvt"):
        for pad in (0, 4):
            P.append(spin(op, 2 * n_cu, two, pad))
    for op in ("ldsr", "ldsw", "ldsr_random", "ldsw_random"):
        for pad in (0, 4):
            P.append(spin(op, 2 * n_cu, two, pad, iters=30000))
    for cus in (n_cu // 4, n_cu // 2):
        P.append(spin("fma64", cus, one, 0))
    G = 1 << 30
    res["streams"] = [stream("hbm", 16 * G, 16 * G, 0), stream("hbm_random", 16 * G, 16 * G, 0, random_data=1),
                      stream("hbm_8byte_loads", 16 * G, 16 * G, 0, random_data=2), stream("hbm_random_8byte_loads", 16 * G, 16 * G, 0, random_data=3),
                      stream("infinity_cache", 128 << 20, 16 * G, 16 << 20), stream("l2", 8 << 20, 16 * G, 1 << 20),
                      stream("l2_random", 8 << 20, 16 * G, 1 << 20, random_data=1)]
    _, idle2 = powered(lambda: time.sleep(SEC))
    res["idle_after_w"] = idle2
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
