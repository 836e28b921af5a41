"""Synthetic chirp-prefixed frame generator for benchmarks and property tests.

Follows the transmit side of the reference (transmitter.transmit / send_to_stream,
OFDM.py:242-259, 296-343): Gray-mapped payload on the data carriers, Hermitian
symmetric N-bin symbols, IFFT, cyclic prefix, known-symbol pilots before and after
each packet, x2 gain, chirp prefix.  Runs on the GPU with torch (input generation
is not part of the measured receive path).
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import RxConfig, map_bits


def known_time_symbol(cfg: RxConfig, device):
    kn = torch.zeros(cfg.N, dtype=torch.complex128, device=device)
    ks = torch.from_numpy(cfg.known_symbols()).to(device)
    car = torch.arange(1, cfg.K + 1, device=device)
    kn[car] = ks
    kn[cfg.N - car] = torch.conj(ks)
    t = torch.fft.ifft(kn)
    return torch.cat([t[cfg.N - cfg.CP:], t]) if cfg.CP else t


def make_frames(cfg: RxConfig, chirp: np.ndarray, F: int, seed: int, stride: int, gmax: int = 300,
                device="cuda", dtype=torch.float32, noise_sigma: float = 0.0):
    """F distinct frame rows [F, stride]: row f = [g_f zeros | chirp | P pilots | D data | P pilots | zeros].
    Returns (rows, payload_bits uint8 [F, D*C*mu], gaps int64 [F])."""
    assert stride >= gmax + cfg.frame_len
    rs = np.random.RandomState(seed)
    nb = cfg.D * cfg.C * cfg.mu
    payload = rs.randint(0, 2, (F, nb)).astype(np.uint8)
    gaps = rs.randint(0, gmax, F) if gmax > 0 else np.zeros(F, dtype=np.int64)
    bins = np.asarray(cfg.data_bins)
    unused = np.delete(np.arange(1, cfg.K + 1), bins - 1)
    qpsk = np.array([1 + 1j, 1 - 1j, -1 + 1j, -1 - 1j]) / np.sqrt(2)
    fill = qpsk[rs.randint(0, 4, len(unused))]                 # one filler vector for all symbols (OFDM.py:210-215)
    sym = map_bits(payload.reshape(F * cfg.D, cfg.C, cfg.mu), cfg.const_points, cfg.const_bits)
    X = torch.zeros((F * cfg.D, cfg.N), dtype=torch.complex128, device=device)
    tsym = torch.from_numpy(sym).to(device)
    tb = torch.from_numpy(bins).to(device)
    X[:, tb] = tsym
    X[:, cfg.N - tb] = torch.conj(tsym)
    if len(unused):
        tu = torch.from_numpy(unused).to(device)
        tf = torch.from_numpy(fill).to(device)
        X[:, tu] = tf
        X[:, cfg.N - tu] = torch.conj(tf)
    td = torch.fft.ifft(X, dim=1)
    if cfg.CP:
        td = torch.cat([td[:, cfg.N - cfg.CP:], td], dim=1)
    td = td.reshape(F, cfg.D * cfg.S)
    kt = known_time_symbol(cfg, device).repeat(cfg.P)
    body = 2.0 * torch.cat([kt.expand(F, -1), td, kt.expand(F, -1)], dim=1).real
    frame = torch.cat([torch.from_numpy(chirp).to(device).expand(F, -1), body], dim=1)      # [F, frame_len]
    rows = torch.zeros((F, stride), dtype=torch.float64, device=device)
    idx = torch.from_numpy(gaps).to(device).unsqueeze(1) + torch.arange(cfg.frame_len, device=device).unsqueeze(0)
    rows.scatter_(1, idx, frame)
    if noise_sigma > 0:
        g = torch.Generator(device=device).manual_seed(seed + 1)
        rows += noise_sigma * torch.randn(rows.shape, generator=g, device=device, dtype=torch.float64)
    return rows.to(dtype), payload, gaps


def tile_rows(rows: torch.Tensor, F_total: int):
    """[F_total, stride] made of repeats of the distinct rows (built in place, no host copy)."""
    Fd, stride = rows.shape
    out = torch.empty((F_total, stride), dtype=rows.dtype, device=rows.device)
    for s in range(0, F_total, Fd):
        e = min(F_total, s + Fd)
        out[s:e] = rows[: e - s]
    return out
