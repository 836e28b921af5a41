"""gf3_audio_modem_amd -- MI355X-native OFDM receive path for the GF3 audio modem.

Layout
  csrc/      hand-written HIP kernels + the C ABI (include/gf3rx.h) -> lib/libgf3rx.so
  _lib.py    ctypes binding of that ABI (fails loudly when the library is missing)
  engine.py  Engine: torch-tensor front end of the ABI (device memory + streams only)
  OFDM.py    drop-in mirror of the reference's `receiver` class (same names/shapes)
  dist.py    frame sharding across GPUs + the all-gather of packed bits (overlapped per chunk)
"""
from .engine import Engine, RxConfig, qpsk_table, square_qam_table  # noqa: F401

__version__ = "0.1.0"
