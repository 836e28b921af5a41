"""CPU-side checks of the boundary: the library loads, exports every symbol the
header declares, and the host mirror keeps the reference's parameter block."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gf3_audio_modem_amd import _lib, build
    build.build_lib()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "gf3rx.h")).read()
    declared = set(re.findall(r"\b(gf3_[a-z_]+)\s*\(", hdr))
    declared -= {"gf3_status", "gf3_dtype", "gf3_config", "gf3_ctx"}
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"libgf3rx.so lacks {name}"
    assert set(_lib.exported_names()) == declared
    assert lib.gf3_version() == b"0.1.0"


def test_config_struct_matches_header_order():
    from gf3_audio_modem_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gf3rx.h")).read()
    body = hdr[hdr.index("typedef struct {"): hdr.index("} gf3_config;")]
    names = []
    for line in body.splitlines():
        line = line.split("/*")[0].strip()
        if line.endswith(";"):
            decl = line[:-1]
            for part in decl.split(",") if not decl.startswith("const") else [decl]:
                names.append(part.replace("*", " ").split()[-1])
    assert [f[0] for f in _lib.Gf3Config._fields_] == names


def test_engine_refuses_without_gpu():
    import torch
    from gf3_audio_modem_amd import Engine, RxConfig
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = RxConfig(N=1024, CP=128, P=2, D=8, data_bins=np.arange(1, 511), known_bits=np.zeros(4096, np.uint8))
    with pytest.raises(Exception, match="no GPU|no CPU fallback"):
        Engine(cfg)


def test_facade_parameter_block_matches_reference_defaults():
    from gf3_audio_modem_amd.OFDM import receiver
    from tests.util import load
    rx = receiver(mode="A2", encoding="XOR")
    assert (rx.ofdm_symbol_size, rx.K, rx.cp_length, rx.no_pilots, rx.packet_length) == (4096, 2047, 224, 20, 180)
    assert rx.chirp_length == 5 * (4096 + 224) and rx.data_carriers_per_symbol == 1400
    assert rx.data_bits_per_symbol == 2800 and rx.bits_per_symbol == 4094 and rx.mu == 2
    g = load("g6_realrec")
    assert np.array_equal(rx.known_sequence, g["known_bits"])          # what the reference read from random_bits.txt
    assert "Cyclic prefix length:               224" in repr(rx)
    for mode, cp in (("B1", 704), ("C3", 1184)):
        assert receiver(mode=mode).cp_length == cp
    with pytest.raises(KeyError):
        receiver(mode="Z9")


def test_tables_match_oracle():
    from gf3_audio_modem_amd import qpsk_table, square_qam_table
    from oracle import gf3_oracle as orc
    assert np.array_equal(qpsk_table()[0], orc.qpsk_table()[0])
    for mu in (4, 6):
        a, b = square_qam_table(mu), orc.square_qam_table(mu)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_load_save_file_framing(tmp_path, monkeypatch, capsys):
    """File framing of OFDM.py:756-794 (host I/O, no GPU): name\\0size\\0 + bytes, and its inverse."""
    from gf3_audio_modem_amd.OFDM import load_file, save_file
    monkeypatch.chdir(tmp_path)
    (tmp_path / "input_files").mkdir()
    blob = np.arange(300, dtype=np.uint8)
    blob.tofile(tmp_path / "input_files" / "abc.bmp")
    bits = load_file("abc.bmp")
    hdr = b"abc.bmp\x00300\x00"
    assert np.array_equal(np.packbits(bits)[: len(hdr)], np.frombuffer(hdr, dtype=np.uint8))
    assert len(bits) == 8 * (len(hdr) + 300)
    name, data = save_file(np.concatenate([bits, np.ones(37, dtype=np.uint8)]))      # trailing padding is ignored
    assert name == "abc.bmp" and np.array_equal(data, blob)
    assert (tmp_path / "output_files" / "abc_received.bmp").exists()
