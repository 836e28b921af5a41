#!/usr/bin/env python3
"""Turn the scratch output of tools/profile_round.sh (gpurun_out/round/) into the tracked files under profiles/.

    python tools/collect_profiles.py [--tag r01_final]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu`),
profiles/<tag>_bench.json (the bench line of the same run set), profiles/<tag>_config3.json / _config5.log when
present, and profiles/traffic_current.json (HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes), which
bench.py reads for `roofline.traffic`.

Counter handling follows MI355X_MICROARCH.md (HBM / rocprofv3 section): separate --pmc passes; values are KB;
on gfx950 FETCH_SIZE reports about half of coalesced streaming reads, so both counters are calibrated in the same
run set on rfft_kernel<2048,f32> (tools/config5.py), whose byte counts are known exactly.
"""
import argparse
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
    return max(hits, key=os.path.getmtime)


def counter_mean(sub, counter, kernel_prefix, grid=None):
    """mean Counter_Value over the dispatches of one kernel (optionally of one grid size)"""
    vals = []
    with open(newest(f"{sub}/**/*_counter_collection.csv")) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter or not row["Kernel_Name"].startswith(kernel_prefix):
                continue
            if grid is not None and int(row["Grid_Size"]) != grid:
                continue
            vals.append(float(row["Counter_Value"]))
    if not vals:
        sys.exit(f"no {counter} rows for {kernel_prefix} in {sub}")
    return sum(vals) / len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01_final")
    args = ap.parse_args()
    os.makedirs(DST, exist_ok=True)

    shutil.copy(newest("trace/**/*_kernel_stats.csv"), os.path.join(DST, f"{args.tag}_kernel_stats.csv"))
    bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(DST, f"{args.tag}_bench.json"), "w"), indent=1)
    for name in ("config3.json", "config5.log"):
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, f"{args.tag}_{name}"))

    F = bench["config"]["frames_per_gpu"]
    # calibration kernel: tools/config5.py's N=4096 case = rfft_kernel<2048, f32>, 4096 transforms:
    # reads 2^24 f32 samples, writes 4096 x 2049 complex128 bins
    cal_read, cal_write = (1 << 24) * 4, 4096 * 2049 * 16
    cal_f = counter_mean("cal_fetch", "FETCH_SIZE", "void rfft_kernel<2048, 1>") * 1024 / cal_read
    cal_w = counter_mean("cal_write", "WRITE_SIZE", "void rfft_kernel<2048, 1>") * 1024 / cal_write
    raw, corr = {}, {}
    algo = {"demod_kernel": bench["roofline"]["algorithmic_bytes_per_launch"],
            "corr_kernel": bench["roofline_sync"]["algorithmic_bytes_per_launch"]}
    for k in ("demod_kernel", "corr_kernel"):
        f = counter_mean("fetch", "FETCH_SIZE", f"void {k}<")
        w = counter_mean("write", "WRITE_SIZE", f"void {k}<")
        raw[k] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
        corr[k] = {"read": f * 1024 / cal_f, "write": w * 1024 / cal_w, "algorithmic": algo[k]}
    out = {
        "_comment": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; "
                    "python3 bench.py --steps 2 --warmup 1 --no-cpu). Counters are in KB. gfx950 correction "
                    "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of coalesced streaming reads; "
                    "calibrated in the same run set on rfft_kernel<2048,f32> (tools/config5.py), whose byte counts "
                    "are known. Produced by tools/collect_profiles.py.",
        "calibration": {"fetch_reported_over_actual": cal_f, "write_reported_over_actual": cal_w},
        "raw_KB": raw,
        f"corrected_bytes_per_launch_F{F}": corr,
        "demod_kernel_bytes_per_launch_at_F": {str(F): corr["demod_kernel"]["read"] + corr["demod_kernel"]["write"]},
    }
    json.dump(out, open(os.path.join(DST, "traffic_current.json"), "w"), indent=1)
    d = corr["demod_kernel"]
    print(f"demod: read {d['read'] / 1e9:.3f} GB + write {d['write'] / 1e9:.3f} GB vs algorithmic {d['algorithmic'] / 1e9:.3f} GB")
    c = corr["corr_kernel"]
    print(f"sync : read {c['read'] / 1e9:.3f} GB + write {c['write'] / 1e9:.3f} GB vs algorithmic {c['algorithmic'] / 1e9:.3f} GB")
    print(f"calibration: fetch x{cal_f:.4f}, write x{cal_w:.4f}")


if __name__ == "__main__":
    main()
