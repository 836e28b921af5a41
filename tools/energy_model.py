#!/usr/bin/env python3
"""Energy budget of the two hot kernels, rebuilt from committed measurements only:

    profiles/r03_energy_budget.json    tools/ubench/energy_budget.py: package power of instruction streams at several issue
                                       densities and of streaming reads -> energy per operation as the SLOPE of power over rate,
                                       and the power of a grid that is resident but issues nothing as the intercept
    profiles/r03_instruction_mix.json  tools/ab/pmc_mix.sh: wave-instructions per launch by class (SQ_INSTS_* counters)
    profiles/r03_kernel_power.json     tools/kernel_power.py: launch time and sustained package power of each kernel alone
    profiles/traffic_current.json      HBM bytes per launch of demod_kernel (FETCH_SIZE / WRITE_SIZE passes)

    python tools/energy_model.py [--md]

Per kernel:  E_launch = P_resident * t  +  sum_class n_class * 64 lanes * e_class  +  bytes * e_hbm   against   P_measured * t.
"""
import argparse
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)


def slopes(eb):
    """energy per lane-operation (pJ) by op, and the resident-grid power (W)"""
    by = {}
    for p in eb["points"]:
        if p["clock_MHz"] < 2300.0 and p["op"] not in ("sleep", "pad_only"):
            continue                                        # clock-throttled point (lower voltage): not on the 2.4 GHz line
        by.setdefault(p["op"], []).append((p["lane_ops_per_s"], p["package_w"], p["workgroups"], p["lds_bytes"]))
    resident = float(np.mean([w for _, w, _, _ in by["sleep"]] + [w for _, w, _, _ in by["pad_only"]]))
    e = {}
    for op, pts in by.items():
        if op in ("sleep", "pad_only"):
            continue
        full = [(r, w) for r, w, wg, _ in pts if wg >= 2 * eb["n_cu"] or op.startswith("lds")]
        full = full or [(r, w) for r, w, _, _ in pts]
        # least squares through the resident power: P = resident + rate * e; points within 5 % of the power cap or
        # visibly clock-throttled are left out by the caller's choice of densities
        r = np.array([x for x, _ in full]); w = np.array([y for _, y in full])
        e[op] = float(np.sum(r * (w - resident)) / np.sum(r * r)) * 1e12
    if "salu" in e:
        e["salu"] *= 64.0                                   # the driver counts 256 "lanes" per workgroup: a scalar instruction runs once per wave
    st = {s["stream"]: (s["package_w"] - resident) / s["bytes_per_s"] * 1e12 for s in eb["streams"]}
    return resident, e, st


def budget(mix, t_s, hbm_bytes, resident, e, st, random_data):
    """rows (label, wave-instructions, pJ per lane-operation, joule) with the prices measured on operands that carry
    random bits (random_data) or on the structured operands of the plain loops (a handful of values shared by all lanes)"""
    pick = lambda k, alt: e.get(k + "_random", e.get(k, alt)) if random_data else e.get(k, alt)
    other = mix["SQ_INSTS_VALU"] - sum(mix.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                                                "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32",
                                                                "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_FMA_F32"))
    lds_e = 0.5 * (pick("ldsr", 52.0) + pick("ldsw", 61.0))
    rows = [
        ("resident grid (clocks, leakage, idle HBM): %.0f W x t" % resident, None, None, resident * t_s),
        ("v_fma_f64", mix["SQ_INSTS_VALU_FMA_F64"], pick("fma64", 30.0), None),
        ("v_add_f64", mix["SQ_INSTS_VALU_ADD_F64"], pick("add64", 20.0), None),
        ("v_mul_f64", mix["SQ_INSTS_VALU_MUL_F64"], pick("mul64", 26.0), None),
        ("v_rcp / v_rsq / v_sqrt f64 (priced as 4 fma)", mix["SQ_INSTS_VALU_TRANS_F64"], 4 * pick("fma64", 30.0), None),
        ("f32 <-> f64 conversions", mix["SQ_INSTS_VALU_CVT"], e.get("cvt", 10.0), None),
        ("32-bit integer add / mul", mix["SQ_INSTS_VALU_INT32"], e.get("addu32", 15.0), None),
        ("64-bit integer", mix["SQ_INSTS_VALU_INT64"], 1.3 * e.get("addu32", 15.0), None),
        ("other VALU (moves, selects, logic, shifts, compares; mean of v_mov and v_cndmask)", other, 0.5 * (e.get("mov", 7.5) + e.get("cndmask", 12.5)), None),
        ("LDS instructions (mean of 16-byte read and write)", mix["SQ_INSTS_LDS"], lds_e, None),
        ("scalar ALU", mix["SQ_INSTS_SALU"], e.get("salu", 55.0) / 64.0, None),
    ]
    out, total = [], 0.0
    for label, n, pj, joule in rows:
        if joule is None:
            joule = n * 64.0 * pj * 1e-12
        total += joule
        out.append((label, n, pj, joule))
    e_hbm = st.get("hbm_random", st["hbm"]) if random_data else st["hbm"]
    j = hbm_bytes * e_hbm * 1e-12
    out.append(("HBM -> registers: %.2f GB (streaming-read microbenchmark: L2 + fabric + HBM)" % (hbm_bytes / 1e9), hbm_bytes / 64.0, e_hbm, j))
    total += j
    return out, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", action="store_true", help="markdown table")
    args = ap.parse_args()
    eb = json.load(open(P("r03_energy_budget.json")))
    mixfile = P("instruction_mix_current.json") if os.path.exists(P("instruction_mix_current.json")) else P("r03_instruction_mix.json")
    mix = json.load(open(mixfile))["kernels"]
    key = lambda prefix, grid: next(k for k in mix if k.startswith(prefix) and k.endswith(f"grid={grid}"))
    kp = json.load(open(P("r03_kernel_power.json")))
    tr = json.load(open(P("traffic_current.json")))
    resident, e, st = slopes(eb)
    print("resident-grid power %.0f W (idle %.0f W); pJ per lane-operation: %s" % (resident, eb["idle_w"], {k: round(v, 1) for k, v in sorted(e.items())}))
    print("pJ per byte streamed: %s" % {k: round(v, 1) for k, v in st.items()})
    F = 65536
    demod_bytes = list(tr["demod_kernel_bytes_per_launch_at_F"].values())[0]
    corr_bytes = F * (4 * (23040 + 320 - 1) + 8)                  # algorithmic = measured to 0.5 % (profiles/r03_pmc.json)
    for kname, mkey, pkey, nbytes in (("demod_kernel<2048,f32,MODE_QPSK>", key("demod_kernel<2048, 1, false, 2", 16777216), "demod_kernel", demod_bytes),
                                      ("corr_kernel<1024,f32>", key("corr_kernel<1024, 1>", 8388608), "corr_kernel", corr_bytes)):
        t_s, p_w = kp[pkey]["ms_per_launch"] * 1e-3, kp[pkey]["package_power_w"]
        measured = p_w * t_s
        lo_rows, lo = budget(mix[mkey], t_s, nbytes, resident, e, st, False)
        hi_rows, hi = budget(mix[mkey], t_s, nbytes, resident, e, st, True)
        print("\n%s: %.3f ms per launch at %.0f W = %.3f J = %.1f uJ per packet" % (kname, kp[pkey]["ms_per_launch"], p_w, measured, measured / F * 1e6))
        if args.md:
            print("| item | wave-instructions per launch | pJ per lane-op (structured ... random operands) | uJ per packet | share of measured |")
            print("|---|---|---|---|---|")
        for (label, n, pj0, j0), (_, _, pj1, j1) in zip(lo_rows, hi_rows):
            cnt = "" if n is None or label.startswith("HBM") else "%.1f M" % (n / 1e6)
            price = "" if pj0 is None else ("%.1f" % pj0 if abs(pj1 - pj0) < 0.05 else "%.1f ... %.1f" % (pj0, pj1)) + (" pJ/B" if label.startswith("HBM") else "")
            uj = "%.1f" % (j0 / F * 1e6) if abs(j1 - j0) < 1e-9 else "%.1f ... %.1f" % (j0 / F * 1e6, j1 / F * 1e6)
            sh = "%.1f %%" % (100 * j0 / measured) if abs(j1 - j0) < 1e-9 else "%.1f ... %.1f %%" % (100 * j0 / measured, 100 * j1 / measured)
            print(("| %s | %s | %s | %s | %s |" if args.md else "  %-92s %10s  %-16s %-14s uJ/packet  %s") % (label, cnt, price, uj, sh))
        tail = "modelled %.1f ... %.1f uJ per packet = %.1f ... %.1f %% of the measured %.1f uJ" % (lo / F * 1e6, hi / F * 1e6, 100 * lo / measured, 100 * hi / measured, measured / F * 1e6)
        print(("| **%s** | | | | |" if args.md else "  %s") % tail)


if __name__ == "__main__":
    main()
