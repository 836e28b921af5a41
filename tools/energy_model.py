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


def budget(name, mix, t_s, p_w, hbm_bytes, resident, e, st):
    fp = lambda k, alt: e.get(k + "_random", e.get(k, alt))
    other = mix["SQ_INSTS_VALU"] - sum(mix.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                                                "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32",
                                                                "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_FMA_F32"))
    lds_e = 0.5 * (e.get("ldsr_random", e["ldsr"]) + e.get("ldsw_random", e["ldsw"]))
    rows = [
        ("resident grid (clocks, leakage, idle HBM): %.0f W x t" % resident, None, None, resident * t_s),
        ("v_fma_f64", mix["SQ_INSTS_VALU_FMA_F64"], fp("fma64", 30.0), None),
        ("v_add_f64", mix["SQ_INSTS_VALU_ADD_F64"], fp("add64", 20.0), None),
        ("v_mul_f64", mix["SQ_INSTS_VALU_MUL_F64"], fp("mul64", 26.0), None),
        ("v_rcp / v_rsq / v_sqrt f64 (priced as 4 fma)", mix["SQ_INSTS_VALU_TRANS_F64"], 4 * fp("fma64", 30.0), None),
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
    e_hbm = st.get("hbm_random", st["hbm"])
    j = hbm_bytes * e_hbm * 1e-12
    out.append(("HBM -> registers, %.2f GB at %.0f pJ/B (streaming-read microbenchmark: L2 + fabric + HBM)" % (hbm_bytes / 1e9, e_hbm), None, None, j))
    total += j
    return out, total, p_w * t_s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", action="store_true", help="markdown table")
    args = ap.parse_args()
    eb = json.load(open(P("r03_energy_budget.json")))
    mix = json.load(open(P("r03_instruction_mix.json")))["kernels"]
    kp = json.load(open(P("r03_kernel_power.json")))
    tr = json.load(open(P("traffic_current.json")))
    resident, e, st = slopes(eb)
    print("resident-grid power %.0f W (idle %.0f W); pJ per lane-operation: %s" % (resident, eb["idle_w"], {k: round(v, 1) for k, v in sorted(e.items())}))
    print("pJ per byte streamed: %s" % {k: round(v, 1) for k, v in st.items()})
    F = 65536
    demod_bytes = list(tr["demod_kernel_bytes_per_launch_at_F"].values())[0]
    corr_bytes = F * (4 * (23040 + 320 - 1) + 8)                  # algorithmic = measured to 0.5 % (profiles/r02_pmc.json)
    for kname, mkey, pkey, nbytes in (("demod_kernel<2048,f32,MODE_QPSK>", "demod_kernel<2048, 1, false, 2> grid=16777216", "demod_kernel", demod_bytes),
                                      ("corr_kernel<1024,f32>", "corr_kernel<1024, 1> grid=8388608", "corr_kernel", corr_bytes)):
        rows, total, measured = budget(kname, mix[mkey], kp[pkey]["ms_per_launch"] * 1e-3, kp[pkey]["package_power_w"], nbytes, resident, e, st)
        print("\n%s: %.3f ms per launch at %.0f W = %.3f J = %.1f uJ per packet" % (kname, kp[pkey]["ms_per_launch"], kp[pkey]["package_power_w"], measured, measured / F * 1e6))
        for label, n, pj, joule in rows:
            cnt = "" if n is None else "%8.1f M x 64 x %5.1f pJ" % (n / 1e6, pj)
            line = ("| %s | %s | %.3f J | %.1f uJ | %.1f %% |" if args.md else "  %-100s %-28s %6.3f J  %6.1f uJ/packet  %5.1f %%") % (
                label, cnt, joule, joule / F * 1e6, 100 * joule / measured)
            print(line)
        print("  modelled %.3f J = %.1f %% of the measured %.3f J; unattributed %.1f uJ per packet" % (total, 100 * total / measured, measured, (measured - total) / F * 1e6))


if __name__ == "__main__":
    main()
