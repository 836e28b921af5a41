#!/usr/bin/env python3
"""Turn the scratch output of tools/profile_round.sh (gpurun_out/round/) into the tracked files under profiles/.

    python tools/collect_profiles.py --tag r02

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace of `python3 bench.py --no-cpu --no-power`: the --stats
summary recomputed PER (kernel, grid) from the trace, because one kernel name is launched at several sizes by the bench's
legs and a pooled mean cannot be compared with any one leg's figure),
profiles/<tag>_bench.json (the bench line of the same run set), profiles/<tag>_pmc.json (HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes for every kernel of the bench; issue / wait / LDS shares and the clock of the hot
kernels from the SQ and GRBM passes), profiles/<tag>_instruction_mix.json (SQ_INSTS_* by class for the two hot kernels) and
the two files bench.py reads -- profiles/traffic_current.json (`roofline.traffic`) and profiles/instruction_mix_current.json
(`roofline.power`'s model) -- both STAMPED with the source_sha16 of the library the passes ran on; bench.py quotes them
only when the library it has loaded carries the same stamp.

Counter handling follows MI355X_MICROARCH.md (HBM / rocprofv3 section): separate --pmc passes; values are KB;
on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so both counters are calibrated in
the same run on rfft_kernel<2048,f32> over 2^28 samples (bench.py's config-5 leg), whose byte counts are known exactly
(1 GiB read, 65 536 x 2 049 complex128 bins written).  SQ_* counters are quad-cycles (4 shader clocks).
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round")
DST = os.path.join(ROOT, "profiles")


def newest(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        sys.exit(f"missing {pattern} under {SRC}")
    return max(hits, key=os.path.getsize)


def counters(sub):
    """{(kernel, grid): {counter: (mean value, mean duration us, launches)}} over every process of the pass"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "at::native" in k or "rocclr" in k:
                    continue
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                acc[(k, int(row["Grid_Size"]))][row["Counter_Name"]].append((float(row["Counter_Value"]), dur))
    out = {}
    for key, cs in acc.items():
        out[key] = {c: (sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v), len(v)) for c, v in cs.items()}
    return out


def kernel_stats_per_grid(dst):
    """rocprofv3's --stats table, one row per (kernel name, grid size) instead of one per name"""
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(SRC, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                rows[(r["Kernel_Name"], grid, int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in rows.values()) or 1
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "GridThreads", "WorkgroupThreads", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for (name, grid, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            mean = sum(v) / len(v)
            sd = (sum((x - mean) ** 2 for x in v) / len(v)) ** 0.5
            w.writerow([name, grid, wg, len(v), sum(v), round(mean, 3), round(100.0 * sum(v) / total, 2), min(v), max(v), round(sd, 3)])


def instruction_mix(tag, sha16):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("mixa", "mixb"):
        for f in glob.glob(os.path.join(SRC, sub, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith(("demod_kernel", "corr_kernel")):
                    acc[f"{k} grid={int(row['Grid_Size'])}"][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if not acc:
        return
    out = {"_comment": "rocprofv3 --pmc passes (two, 8 SQ counters each) of `python3 bench.py --steps 2 --warmup 1 --no-cpu --no-power` (hot-path "
                       "legs only): wave-instructions per launch by class, mean over the launches of the run; tools/profile_round.sh",
           "source_sha16": sha16,
           "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}}
    json.dump(out, open(os.path.join(DST, f"{tag}_instruction_mix.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(DST, "instruction_mix_current.json"), "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r04")
    args = ap.parse_args()
    os.makedirs(DST, exist_ok=True)
    kernel_stats_per_grid(os.path.join(DST, f"{args.tag}_kernel_stats.csv"))
    bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(DST, f"{args.tag}_bench.json"), "w"), indent=1)
    sha16 = bench["library"]["source_sha16"]
    instruction_mix(args.tag, sha16)

    fetch, write, sq, grbm = counters("fetch"), counters("write"), counters("sq"), counters("grbm")
    big = 16777216                                            # 65 536 workgroups of 256 threads
    cal_read, cal_write = (1 << 28) * 4, 65536 * 2049 * 16
    cal_f = fetch[("rfft_kernel<2048, 1>", big)]["FETCH_SIZE"][0] * 1024 / cal_read
    cal_w = write[("rfft_kernel<2048, 1>", big)]["WRITE_SIZE"][0] * 1024 / cal_write
    F = bench["config"]["frames_per_gpu"]
    algo = {"demod_kernel<2048, 1, false, 2, 0>": bench["roofline"]["algorithmic_bytes_per_launch"],
            "corr_kernel<1024, 1>": bench["roofline_sync"]["algorithmic_bytes_per_launch"]}
    for k, leg in (("demod_kernel<2048, 1, false, 1, 0>", "roofline_demod_16qam"), ("soft_demap_bin_kernel<3, 3>", "roofline_soft_demap")):
        if leg in bench:
            algo[k] = bench[leg]["algorithmic_bytes_per_launch"]
    for N, leg in bench.get("roofline_rfft", {}).items():
        algo[f"rfft_kernel<{int(N[1:]) // 2}, 1>"] = leg["algorithmic_bytes_per_launch"]
    if "roofline_stream_sync" in bench:
        algo["scr_ring_kernel<1>"] = algo["scr_ols_kernel<1>"] = bench["roofline_stream_sync"]["hbm"]["algorithmic_bytes_per_call"]
    traffic = {}
    for (k, grid), cs in sorted(fetch.items()):
        if (k, grid) not in write or cs["FETCH_SIZE"][1] < 100.0:       # launches under 0.1 ms: set-up transforms, tiny lists
            continue
        f_kb, w_kb = cs["FETCH_SIZE"][0], write[(k, grid)]["WRITE_SIZE"][0]
        ent = {"grid_threads": grid, "launches_profiled": cs["FETCH_SIZE"][2], "avg_us_under_pmc": round(cs["FETCH_SIZE"][1], 1),
               "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
               "read_bytes_corrected": f_kb * 1024 / cal_f, "write_bytes_corrected": w_kb * 1024 / cal_w}
        ent["hbm_bytes"] = ent["read_bytes_corrected"] + ent["write_bytes_corrected"]
        if k in algo and (grid == big or k.startswith(("scr_", "soft_")) or (k.startswith("corr_") and grid == big // 2)):
            ent["algorithmic_bytes"] = algo[k]
            ent["traffic_over_algorithmic"] = ent["hbm_bytes"] / algo[k]
        traffic[f"{k} grid={grid}"] = ent
    shares = {}
    for (k, grid), cs in sorted(sq.items()):
        if "SQ_WAVE_CYCLES" not in cs or (cs["SQ_WAVE_CYCLES"][1] < 100.0 and not k.startswith("scr_")):
            continue
        wc = cs["SQ_WAVE_CYCLES"][0]
        ent = {"grid_threads": grid, "avg_us_under_pmc": round(cs["SQ_WAVE_CYCLES"][1], 1),
               "valu_wave_instructions": cs["SQ_INSTS_VALU"][0], "lds_wave_instructions": cs["SQ_INSTS_LDS"][0],
               "share_of_wave_cycles": {"valu_issue": cs["SQ_ACTIVE_INST_VALU"][0] / wc, "lds_issue": cs["SQ_ACTIVE_INST_LDS"][0] / wc,
                                        "waiting_waitcnt_or_barrier": cs["SQ_WAIT_ANY"][0] / wc, "issue_stalled": cs["SQ_WAIT_INST_ANY"][0] / wc}}
        g = grbm.get((k, grid))
        if g and "GRBM_GUI_ACTIVE" in g:
            ent["clock_GHz_under_pmc"] = g["GRBM_GUI_ACTIVE"][0] / 8 / (g["GRBM_GUI_ACTIVE"][1] * 1e-6) / 1e9
            ent["lds_bank_conflict_share_of_lds_cycles"] = g["SQ_LDS_BANK_CONFLICT"][0] / max(1.0, g["SQ_LDS_IDX_ACTIVE"][0])
        shares[f"{k} grid={grid}"] = ent
    out = {"_comment": "rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 --no-cpu --no-power` (FETCH_SIZE, WRITE_SIZE: full bench; "
                       "SQ and GRBM: hot-path legs only). KB counters; FETCH_SIZE calibrated on rfft_kernel<2048,f32> over 2^28 samples in the "
                       "same run (MI355X_MICROARCH.md: gfx950 reports half of wide streaming reads); narrower access patterns "
                       "(scr_ols_kernel's 8-byte sample pairs) are corrected by the same factor and are upper bounds. Produced by tools/collect_profiles.py.",
           "calibration": {"fetch_reported_over_actual": cal_f, "write_reported_over_actual": cal_w},
           "hbm_traffic_per_launch": traffic, "sq_shares": shares}
    json.dump(out, open(os.path.join(DST, f"{args.tag}_pmc.json"), "w"), indent=1)
    d = traffic[f"demod_kernel<2048, 1, false, 2, 0> grid={big}"]
    cur = {"_comment": f"see {args.tag}_pmc.json", "source_sha16": sha16, "calibration": out["calibration"],
           "demod_kernel_bytes_per_launch_at_F": {str(F): d["hbm_bytes"]}}
    # one screened gf3_sync_stream call on the config-3 stream = one launch of each of these kernels (plus two scans and the
    # suppression walk, which are noise): VALU wave-instructions (SQ pass) and HBM bytes (FETCH / WRITE passes) per call
    call = ("scr_ring_kernel", "scr_ols_kernel", "scr_mlo_kernel", "scr_flag_kernel", "scr_scatter_kernel", "scr_refine_kernel",
            "scr_decide_kernel", "scr_expand_kernel")
    insts = sum(cs["SQ_INSTS_VALU"][0] for (k, grid), cs in sq.items() if k.split("<")[0] in call and "SQ_INSTS_VALU" in cs)
    by = sum(cs["FETCH_SIZE"][0] * 1024 / cal_f + write[(k, grid)]["WRITE_SIZE"][0] * 1024 / cal_w
             for (k, grid), cs in fetch.items() if k.split("<")[0] in call and (k, grid) in write)
    if insts:
        cur["stream_sync_valu_wave_instructions_per_call"] = insts
    if by:
        cur["stream_sync_hbm_bytes_per_call"] = by
    json.dump(cur, open(os.path.join(DST, "traffic_current.json"), "w"), indent=1)
    for k, e in traffic.items():
        if "traffic_over_algorithmic" in e:
            print(f"{k:55s} HBM {e['hbm_bytes'] / 1e9:7.3f} GB = {e['traffic_over_algorithmic']:.3f} x algorithmic")
    for k, e in shares.items():
        s = e["share_of_wave_cycles"]
        print(f"{k:55s} VALU {s['valu_issue']:.2f} LDS {s['lds_issue']:.2f} wait {s['waiting_waitcnt_or_barrier']:.2f} stall {s['issue_stalled']:.2f} clk {e.get('clock_GHz_under_pmc', 0):.2f} GHz")


if __name__ == "__main__":
    main()
