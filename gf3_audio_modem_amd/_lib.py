"""ctypes binding of include/gf3rx.h.  There is no fallback: if libgf3rx.so is
missing or does not load, importing the engine raises."""
import ctypes as C
import os

from .build import lib_path

GF3_OK, GF3_EINVAL, GF3_EHIP, GF3_ENOMEM, GF3_ERANGE, GF3_ENODETECT = 0, -1, -2, -3, -4, -5
DT_F64, DT_F32, DT_I16, DT_U8 = 0, 1, 2, 3

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)


class Gf3Config(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("CP", C.c_int32), ("P", C.c_int32), ("D", C.c_int32), ("Lc", C.c_int32),
        ("fs", C.c_double), ("f0", C.c_double), ("f1", C.c_double), ("thresh", C.c_double),
        ("fit_lo", C.c_int32), ("fit_hi", C.c_int32), ("mu", C.c_int32), ("M", C.c_int32),
        ("const_re", c_double_p), ("const_im", c_double_p), ("const_bits", C.POINTER(C.c_uint8)),
        ("known_re", c_double_p), ("known_im", c_double_p),
        ("data_bins", C.POINTER(C.c_int32)), ("C", C.c_int32),
        ("in_dtype", C.c_int32), ("max_window", C.c_int32),
    ]


_SIGS = {
    "gf3_version": (C.c_char_p, []),
    "gf3_source_hash": (C.c_char_p, []),
    "gf3_ctx_create": (C.c_int, [C.POINTER(Gf3Config), C.POINTER(C.c_void_p)]),
    "gf3_ctx_destroy": (None, [C.c_void_p]),
    "gf3_last_error": (C.c_char_p, [C.c_void_p]),
    "gf3_clear_runtime_error": (C.c_int, []),
    "gf3_debug_set_stamps": (None, [C.c_void_p, C.c_void_p]),
    "gf3_bytes_per_frame": (C.c_int32, [C.c_void_p]),
    "gf3_sync_max_window": (C.c_int32, [C.c_void_p]),
    "gf3_sync_stream_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "gf3_chirp_replica": (C.c_int, [C.c_void_p, c_double_p]),
    "gf3_rfft_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gf3_demod_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gf3_demod_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "gf3_demod_split_plan": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gf3_demod_frames_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p]),
    "gf3_equalise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gf3_sync_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "gf3_sync_frames_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "gf3_sync_frames_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gf3_debug_frames_screen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gf3_sync_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, c_i64_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "gf3_sync_stream_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, c_i64_p,
                                     C.c_void_p, C.c_void_p, C.c_int32, c_i64_p, C.c_void_p]),
    "gf3_sync_chunk_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "gf3_sync_chunk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, c_i64_p, c_double_p, C.c_void_p, C.c_void_p]),
    "gf3_sync_decide_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "gf3_sync_decide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                  c_i64_p, C.c_void_p, C.c_void_p]),
    "gf3_sync_stream_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "gf3_sync_stream_info": (C.c_int, [C.c_void_p, c_i64_p]),
    "gf3_debug_stream_screen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "gf3_known_h_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "gf3_equalise_known_h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gf3_tx_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                C.c_int32, C.c_void_p]),
    "gf3_schmidl_cox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "gf3_unpack_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gf3_demap_hard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gf3_soft_demap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]),
}

_lib = None


def exported_names():
    return sorted(_SIGS)


def load():
    """dlopen libgf3rx.so and attach the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    from . import build as _build
    if not os.environ.get("GF3_LIB") and _build.stale():
        # missing, or built from other sources than csrc/*.hip, csrc/*.h and include/gf3rx.h as they are now
        # (the library carries the SHA-256 of what it was built from): compile it here if the ROCm toolchain is present
        # (same gfx950 build as __graft_entry__.build(), temp file + rename under a lock); otherwise fail loudly --
        # kernels that do not match the source must not run, and there is no other implementation to fall back to
        if _build.have_compiler():
            _build.build_lib()
        elif os.path.exists(path):
            have = _build.built_hash()
            why = ("carries no source stamp (built by hand or by an older checkout)" if have is None
                   else f"was built from other sources (stamp {have[:16]}..., sources now {_build.source_hash()[:16]}...)")
            raise ImportError(f"{path} {why} and no hipcc is available to rebuild it; "
                              "set GF3_LIB=<path> to load a library of your own choosing")
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is not built. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the receive path.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)           # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def build_id():
    """(library version string, first 16 hex digits of the source hash it was built from) for bench records."""
    from . import build as _build
    lib = load()
    return lib.gf3_version().decode(), lib.gf3_source_hash().decode()[:16]
