"""Build libgf3rx.so in-tree with hipcc for gfx950 (no JIT cache: the built
library travels to the GPU box with the repo snapshot)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "gf3rx.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "gf3rx_device.h"), os.path.join(ROOT, "include", "gf3rx.h")]
LIB = os.path.join(HERE, "lib", "libgf3rx.so")


def lib_path():
    """In-tree library; GF3_LIB=/path/to/other.so selects another build (A/B timing, diagnostic stamp builds)."""
    return os.environ.get("GF3_LIB") or LIB


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-o", LIB] + SRC
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
