"""Build libgf3rx.so in-tree with hipcc for gfx950 (no JIT cache: the built
library travels to the GPU box with the repo snapshot).

Staleness is decided by content, not by time stamps (a snapshot copy does not keep them): the build writes the
SHA-256 of its sources next to the library, and `stale()` compares it with the sources as they are now.  Builds
go to a temporary file that is renamed into place under a file lock, so N ranks that all find the library stale
compile once and never dlopen a half-written file."""
import fcntl
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "gf3rx.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "gf3rx_device.h"), os.path.join(HERE, "csrc", "gf3rx_screen.h"),
              os.path.join(ROOT, "include", "gf3rx.h")]
LIB = os.path.join(HERE, "lib", "libgf3rx.so")
STAMP = LIB + ".srchash"
# -fno-slp-vectorize: LLVM otherwise pairs the screening kernel's fp32 complex arithmetic into v_pk_* instructions,
# which issue at half rate on gfx950 and need register shuffles (scr_ols_kernel 3.0 -> 2.6 ms without them; the fp64
# kernels, which have no packed form, are unchanged)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize"]


def lib_path():
    """In-tree library; GF3_LIB=/path/to/other.so selects another build (A/B timing, diagnostic stamp builds)."""
    return os.environ.get("GF3_LIB") or LIB


def source_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        if os.path.exists(d):
            h.update(os.path.basename(d).encode())
            h.update(open(d, "rb").read())
    return h.hexdigest()


def built_hash():
    try:
        return open(STAMP).read().strip()
    except OSError:
        return None


def stale():
    return not os.path.exists(LIB) or built_hash() != source_hash()


def have_compiler():
    return bool(shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))


def build_lib(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(LIB), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():                  # another process built it while we waited for the lock
                return LIB
            want = source_hash()
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-o", tmp] + SRC
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB)
            with open(STAMP + ".tmp", "w") as fh:
                fh.write(want + "\n")
            os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
