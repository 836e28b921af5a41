"""Build libgf3rx.so in-tree with hipcc for gfx950 (no JIT cache: the built
library travels to the GPU box with the repo snapshot).

The library is made of per-family translation units (csrc/gf3rx_*.hip, see csrc/gf3rx_host.h) that are compiled IN
PARALLEL into objects and linked: a cold build is bounded by the slowest unit instead of the sum, and a rebuild after an
edit recompiles only the units whose inputs changed -- every object's file name carries the SHA-256 of its own source,
of every header and of the flags, so an object is reused exactly when it is still valid (objects travel to the GPU box
with the snapshot too, where a stale library is then a link, not a two-minute compile, inside the first import).

Staleness of the LIBRARY is decided by content, not by time stamps (a snapshot copy does not keep them): the build
compiles the SHA-256 of all sources and flags INTO the library (`gf3_source_hash()`, csrc/gf3rx_stamp.cpp; no side file
that packaging could lose), and `stale()` compares it with the sources as they are now, reading the stamp out of the
file without loading it.  Builds go to a temporary file that is renamed into place under a file lock, so N ranks that
all find the library stale compile once and never dlopen a half-written file."""
import fcntl
import glob
import hashlib
import os
import re
import shutil
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# translation units, slowest first (the pool starts them in this order)
UNITS = ["gf3rx_demod_full", "gf3rx_dsplit_full", "gf3rx_demod_scan", "gf3rx_dsplit_scan", "gf3rx_demod_qpsk", "gf3rx_dsplit_qpsk",
         "gf3rx_screen", "gf3rx_corr", "gf3rx_demod_split", "gf3rx_fscreen",
         "gf3rx_fft", "gf3rx_sync", "gf3rx_abi"]
SRC = [os.path.join(CSRC, u + ".hip") for u in UNITS]
STAMP_SRC = os.path.join(CSRC, "gf3rx_stamp.cpp")
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "gf3rx.h")]
DEPS = SRC + [STAMP_SRC] + HEADERS
LIB = os.path.join(HERE, "lib", "libgf3rx.so")
OBJ = os.path.join(HERE, "lib", "obj")
MARKER = b"GF3_SRC_HASH="
# -fno-slp-vectorize: LLVM otherwise pairs the screening kernel's fp32 complex arithmetic into v_pk_* instructions,
# which issue at half rate on gfx950 and need register shuffles (scr_ols_kernel 3.0 -> 2.6 ms without them; the fp64
# kernels, which have no packed form, are unchanged)
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]
EXTRA = os.environ.get("GF3_EXTRA_CFLAGS", "").split()      # diagnostic builds (-DGF3_STAMPS, -DGF3_DEV_BUILD ...): part of every hash
FLAGS = CFLAGS + LDFLAGS + EXTRA                            # (what the library's stamp covers)


def lib_path():
    """In-tree library; GF3_LIB=/path/to/other.so selects another build (A/B timing, diagnostic stamp builds)."""
    return os.environ.get("GF3_LIB") or LIB


def _digest(paths, extra=""):
    h = hashlib.sha256((" ".join(FLAGS) + extra).encode())
    for d in paths:
        if os.path.exists(d):
            h.update(os.path.basename(d).encode())
            h.update(open(d, "rb").read())
    return h.hexdigest()


def source_hash():
    return _digest(DEPS)


def unit_hash(src):
    """What an object depends on: its own source, every header, the flags."""
    return _digest([src] + HEADERS)


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


def _compile(hipcc, src, obj, defines, verbose):
    tmp = f"{obj}.tmp.{os.getpid()}"
    cmd = [hipcc] + CFLAGS + EXTRA + defines + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + r.stdout + r.stderr)
    os.replace(tmp, obj)
    return time.time() - t0


def build_lib(force=False, verbose=False, out=None, jobs=None):
    """Compile what is out of date and link.  out: write the library there instead of the in-tree path (A/B builds with
    GF3_EXTRA_CFLAGS; objects are keyed by the flags, so they do not collide with the product's)."""
    target = out or LIB
    if not out and not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(os.path.dirname(LIB), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not out and not force and not stale():      # another process built it while we waited for the lock
                return LIB
            want = source_hash()
            todo, objs = [], []
            for src in SRC:
                stem = os.path.splitext(os.path.basename(src))[0]
                obj = os.path.join(OBJ, f"{stem}.{unit_hash(src)[:16]}.o")
                objs.append(obj)
                if force or not os.path.exists(obj):
                    todo.append((src, obj, []))
            stamp_obj = os.path.join(OBJ, f"gf3rx_stamp.{want[:16]}.o")
            objs.append(stamp_obj)
            if force or not os.path.exists(stamp_obj):
                todo.append((STAMP_SRC, stamp_obj, [f'-DGF3_SRC_HASH="{want}"']))
            t0 = time.time()
            workers = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 4)))
            def one(t):
                try:
                    return _compile(hipcc, t[0], t[1], t[2], verbose)
                except RuntimeError as e:                   # (let the other units finish: one run reports every failure)
                    return e
            with ThreadPoolExecutor(workers) as pool:
                took = list(pool.map(one, todo))
            errors = [t for t in took if isinstance(t, RuntimeError)]
            if errors:
                raise RuntimeError("\n".join(str(e) for e in errors))
            if verbose:
                for (src, _, _), s in zip(todo, took):
                    print(f"  {os.path.basename(src):28s} {s:6.1f} s")
                print(f"compiled {len(todo)} of {len(objs)} units in {time.time() - t0:.1f} s", flush=True)
            tmp = f"{target}.tmp.{os.getpid()}"
            cmd = [hipcc] + LDFLAGS + ["-o", tmp] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, target)
            # objects of older sources are of no further use
            keep = set(objs)
            if not EXTRA:
                for old in glob.glob(os.path.join(OBJ, "*.o")):
                    if old not in keep and time.time() - os.path.getmtime(old) > 6 * 3600:
                        os.remove(old)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return target


if __name__ == "__main__":
    import sys
    print(build_lib(force="--force" in sys.argv, verbose=True))
