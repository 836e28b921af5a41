"""The C ABI from C: tests/c_client/gf3_c_client.c -- plain C99 against include/gf3rx.h and the HIP runtime's C API, no
Python, no C++ -- runs receiver.receive()'s path (chirp sync, demodulation, PS + XOR decode).  CPU: the header is valid
C99 and the client compiles and links against the built library.  GPU: it decodes reference fixtures to the reference's bits."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests.util import load, modeA2_params, params_of, unpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client", "gf3_c_client.c")
LIBDIR = os.path.join(ROOT, "gf3_audio_modem_amd", "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(out):
    from gf3_audio_modem_amd import build
    build.build_lib()
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    cmd = [cc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROCM, "include"), SRC, "-o", out, "-L" + LIBDIR, "-lgf3rx", "-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_header_is_c99_and_the_c_client_links(tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "gf3rx.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = _build(str(tmp_path / "gf3_c_client"))
    assert os.path.getsize(exe) > 0


def _write_case(path, p, samples, dtype_code, xor):
    kn = p.known_symbols() if hasattr(p, "known_symbols") else None
    from gf3_audio_modem_amd.engine import map_bits
    K = p.N // 2 - 1
    kn = map_bits(np.asarray(p.known_bits[: K * p.mu]).reshape(K, p.mu), p.const_points, p.const_bits)
    n = len(samples)
    with open(path, "wb") as f:
        f.write(struct.pack("<12i", p.N, p.CP, p.P, p.D, 0, p.mu, len(p.const_points), p.C, dtype_code, n & 0xFFFFFFFF if n < 2 ** 31 else n - 2 ** 32, n >> 32, int(xor)))
        f.write(struct.pack("<4d", 48000.0, 0.0, 8000.0, 0.4))
        f.write(struct.pack("<2i", p.fit_lo, p.fit_hi))
        f.write(np.ascontiguousarray(np.real(p.const_points), dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(np.imag(p.const_points), dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(p.const_bits, dtype=np.uint8).tobytes())
        f.write(np.ascontiguousarray(kn.real, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(kn.imag, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(p.data_carriers, dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(p.known_bits[: p.C * p.mu], dtype=np.uint8).tobytes())
        f.write(np.ascontiguousarray(samples).tobytes())


def _run(exe, case, out):
    r = subprocess.run([exe, case, out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    raw = np.fromfile(out, dtype="<i8")
    assert raw[0] == len(raw) - 1
    return raw[1:], r.stdout


@pytest.mark.gpu
def test_c_client_decodes_the_reference_fixtures(tmp_path):
    """g3 (16-QAM through the measured channel, f64 samples, no whitening) and the reference's own recording (8-bit PCM, mode
    A2, XOR decode: the library picks the two-phase demodulation) through the C program: the reference's bits, the recording's
    by SHA-256."""
    import hashlib
    exe = _build(str(tmp_path / "gf3_c_client"))
    g = load("g3_n4096_16qam_gr5")
    p = params_of(g)
    _write_case(str(tmp_path / "g3.bin"), p, g["r"].astype("<f8"), 0, False)
    bits, out = _run(exe, str(tmp_path / "g3.bin"), str(tmp_path / "g3.out"))
    assert np.array_equal(bits, unpack(g)) and "packets" in out
    g = load("g6_realrec")
    p = modeA2_params(g["known_bits"])
    _write_case(str(tmp_path / "g6.bin"), p, g["wav_u8"], 3, True)
    bits, out = _run(exe, str(tmp_path / "g6.bin"), str(tmp_path / "g6.out"))
    assert "two-phase" in out, out
    assert hashlib.sha256(bits.astype(np.uint8).tobytes()).hexdigest() == str(g["sha256_bits"])
