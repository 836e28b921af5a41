#!/usr/bin/env python3
"""Tabulate hipcc's -Rpass-analysis=kernel-resource-usage remarks (registers, spills, scratch, occupancy, LDS)
per kernel.   hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> res.txt ; python tools/resource_usage.py res.txt
With no argument: compiles the library's translation units itself (--unit=NAME for one of them; --dev for the quick
N=4096-only GF3_DEV_BUILD)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(txt):
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        rows.append((g("VGPRs"), g("AGPRs"), g("SGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), name))
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if args:
        txt = open(args[0]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            sys.path.insert(0, ROOT)
            from concurrent.futures import ThreadPoolExecutor
            from gf3_audio_modem_amd.build import CFLAGS, CSRC, SRC         # exactly the flags the library is built with
            only = [a[7:] for a in sys.argv[1:] if a.startswith("--unit=")]  # e.g. --unit=gf3rx_demod_qpsk
            units = [u for u in SRC if not only or os.path.splitext(os.path.basename(u))[0] in only]
            def one(u):
                cmd = ["/opt/rocm/bin/hipcc"] + CFLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                       "-Rpass-analysis=kernel-resource-usage", "-c", u, "-o", os.path.join(d, os.path.basename(u) + ".o")]
                if "--dev" in sys.argv:
                    cmd.insert(1, "-DGF3_DEV_BUILD")
                return subprocess.run(cmd, capture_output=True, text=True).stderr
            with ThreadPoolExecutor(8) as pool:
                txt = "".join(pool.map(one, units))
    if "--demangle" in sys.argv or True:
        pass
    print(f"{'VGPR':>4} {'AGPR':>4} {'SGPR':>4} {'sSpl':>4} {'vSpl':>4} {'scr':>4} {'occ':>3} {'LDS':>6}  kernel")
    for r in table(txt):
        name = r[8]
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        print(f"{r[0]:>4} {r[1]:>4} {r[2]:>4} {r[3]:>4} {r[4]:>4} {r[5]:>4} {r[6]:>3} {r[7]:>6}  {name[:120]}")
