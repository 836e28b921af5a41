"""Build libgf3rx.so in-tree with hipcc for gfx950 (no JIT cache: the built
library travels to the GPU box with the repo snapshot).

Staleness is decided by content, not by time stamps (a snapshot copy does not keep them): the build compiles the
SHA-256 of its sources and flags INTO the library (`gf3_source_hash()`; no side file that packaging could lose), and
`stale()` compares it with the sources as they are now, reading the stamp out of the file without loading it.  Builds
go to a temporary file that is renamed into place under a file lock, so N ranks that all find the library stale
compile once and never dlopen a half-written file."""
import fcntl
import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "gf3rx.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "gf3rx_device.h"), os.path.join(HERE, "csrc", "gf3rx_screen.h"),
              os.path.join(ROOT, "include", "gf3rx.h")]
LIB = os.path.join(HERE, "lib", "libgf3rx.so")
MARKER = b"GF3_SRC_HASH="
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


def built_hash(path=None):
    """The source hash compiled into the library (None: no library, or one without a stamp -- built by hand / by an
    older checkout)."""
    try:
        blob = open(path or LIB, "rb").read()
    except OSError:
        return None
    m = re.search(MARKER + rb"([0-9a-f]{64})", blob)
    return m.group(1).decode() if m else None


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
            cmd = [hipcc] + FLAGS + [f'-DGF3_SRC_HASH="{want}"', "-I" + os.path.join(ROOT, "include"), "-o", tmp] + SRC
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
