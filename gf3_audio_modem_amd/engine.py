"""Engine: torch-tensor front end of libgf3rx (include/gf3rx.h).

PyTorch is used for device memory, streams and (in dist.py) RCCL only; every
arithmetic step of the receive path runs in the hand-written HIP kernels behind
the C ABI.  There is no CPU path here: without a GPU or without the built
library, constructing an Engine raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib

_TORCH_DT = {_lib.DT_F64: torch.float64, _lib.DT_F32: torch.float32,
             _lib.DT_I16: torch.int16, _lib.DT_U8: torch.uint8}
_DT_OF = {v: k for k, v in _TORCH_DT.items()}


def qpsk_table():
    """QPSK Gray table in the reference's mapping_table order (OFDM.py:72-77)."""
    pts = np.array([(1 + 1j) / np.sqrt(2), (1 - 1j) / np.sqrt(2),
                    (-1 - 1j) / np.sqrt(2), (-1 + 1j) / np.sqrt(2)])
    bits = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.uint8)
    return pts, bits


def square_qam_table(mu):
    """Gray-coded square 2^mu-QAM, unit average energy, points in ascending label
    order (the reference's demap is table-generic, OFDM.py:484-500)."""
    if mu % 2 or mu < 2:
        raise ValueError("Invalid Modulation Type")
    h = mu // 2
    L = 1 << h
    lv = np.zeros(L)
    for i in range(L):
        lv[i ^ (i >> 1)] = 2 * i - (L - 1)
    scale = np.sqrt(2.0 * (L * L - 1) / 3.0)
    pts, bits = [], []
    for lab in range(1 << mu):
        pts.append((lv[lab >> h] + 1j * lv[lab & (L - 1)]) / scale)
        bits.append([(lab >> (mu - 1 - b)) & 1 for b in range(mu)])
    return np.array(pts), np.array(bits, dtype=np.uint8)


def map_bits(bits2d, const_points, const_bits):
    """transmitter.map (OFDM.py:196-197): rows of mu bits -> constellation points."""
    const_bits = np.asarray(const_bits)
    mu = const_bits.shape[1]
    w = 1 << np.arange(mu - 1, -1, -1)
    lut = np.zeros(1 << mu, dtype=complex)
    lut[(const_bits * w).sum(axis=1)] = const_points
    return lut[(np.asarray(bits2d, dtype=np.int64) * w).sum(axis=-1)]


@dataclass
class RxConfig:
    """Python image of gf3_config == the attributes CamG.__init__ sets (OFDM.py:18-101)."""
    N: int = 4096
    CP: int = 224
    P: int = 20
    D: int = 180
    data_bins: np.ndarray = None            # data_carriers (OFDM.py:47)
    const_points: np.ndarray = field(default_factory=lambda: qpsk_table()[0])
    const_bits: np.ndarray = field(default_factory=lambda: qpsk_table()[1])
    known_bits: np.ndarray = None           # known_sequence, >= K*mu bits (OFDM.py:99-101)
    fs: float = 48000.0
    f0: float = 0.0
    f1: float = 8000.0
    thresh: float = 0.4
    fit_lo: int = 500
    fit_hi: int = 1000
    Lc: int = 0
    in_dtype: torch.dtype = torch.float64
    max_window: int = 512

    @property
    def K(self): return self.N // 2 - 1
    @property
    def S(self): return self.N + self.CP
    @property
    def M(self): return 2 * self.P + self.D
    @property
    def mu(self): return int(np.asarray(self.const_bits).shape[1])
    @property
    def C(self): return len(self.data_bins)
    @property
    def chirp_length(self): return self.Lc if self.Lc > 0 else 5 * self.S
    @property
    def frame_len(self): return self.chirp_length + self.M * self.S
    @property
    def bits_per_frame(self): return self.D * self.C * self.mu

    def known_symbols(self):
        kb = np.asarray(self.known_bits[: self.K * self.mu]).reshape(self.K, self.mu)
        return map_bits(kb, self.const_points, self.const_bits)


class Gf3Error(RuntimeError):
    pass


DIRECT_PIECE_BYTES = 128 << 20      # from this size on the runtime pins a pageable source on the fly (its GPU_PINNED_MIN_XFER_SIZE)


def host_pieces(n, chunk_samples, Lc, L):
    """How Engine.receive_host cuts a stream of n samples (chirp length Lc, packet body L = M*S samples): a list of
    pieces, each dict(lo, hi: the NEW samples [lo, hi) it brings; base: stream index of the first sample of its device
    buffer, which starts with the last `carry` = Lc + L + 8 samples of the previous piece; n_buf; g_lo, g_hi: the lags
    [g_lo, g_hi) of the stream's full convolution P (length n + Lc - 1) it owns).  Every lag 1 .. n+Lc-3 -- the p1 of
    every zeros-index of OFDM.py:360 -- is owned by exactly one piece, with its Lc taps and both neighbours inside that
    piece's buffer (or beyond the stream's true ends, where the convolution's zero extension is the reference's own).
    Pure arithmetic: tested on the CPU (tests/test_abi_cpu.py)."""
    carry = Lc + L + 8
    H = max(int(chunk_samples), 2 * carry)
    k = -(-n // H)
    plen = n + Lc - 1
    out = []
    for c in range(k):
        lo, hi = c * H, min(n, (c + 1) * H)
        ce = min(carry, lo)
        out.append(dict(lo=lo, hi=hi, base=lo - ce, n_buf=ce + hi - lo, g_lo=1 if c == 0 else lo - 1,
                        g_hi=plen - 1 if c == k - 1 else hi - 1))
    return out, H, carry


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """One gf3_ctx on one GPU."""

    def __init__(self, cfg: RxConfig, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise Gf3Error("no GPU visible: the gf3rx receive path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if cfg.data_bins is None:
            raise ValueError("RxConfig.data_bins is required")
        if cfg.known_bits is None or len(cfg.known_bits) < cfg.K * cfg.mu:
            raise ValueError("known_bits must hold at least K*mu bits")
        if cfg.in_dtype not in _DT_OF:
            raise ValueError(f"unsupported sample dtype {cfg.in_dtype}")
        self.cfg = cfg
        pts = np.ascontiguousarray(cfg.const_points, dtype=np.complex128)
        self._re = np.ascontiguousarray(pts.real)
        self._im = np.ascontiguousarray(pts.imag)
        self._bits = np.ascontiguousarray(cfg.const_bits, dtype=np.uint8)
        kn = cfg.known_symbols()
        self._kre = np.ascontiguousarray(kn.real)
        self._kim = np.ascontiguousarray(kn.imag)
        self._bins = np.ascontiguousarray(cfg.data_bins, dtype=np.int32)
        g = _lib.Gf3Config(
            N=cfg.N, CP=cfg.CP, P=cfg.P, D=cfg.D, Lc=cfg.Lc, fs=cfg.fs, f0=cfg.f0, f1=cfg.f1,
            thresh=cfg.thresh, fit_lo=cfg.fit_lo, fit_hi=cfg.fit_hi, mu=cfg.mu, M=len(pts),
            const_re=self._re.ctypes.data_as(_lib.c_double_p), const_im=self._im.ctypes.data_as(_lib.c_double_p),
            const_bits=self._bits.ctypes.data_as(C.POINTER(C.c_uint8)),
            known_re=self._kre.ctypes.data_as(_lib.c_double_p), known_im=self._kim.ctypes.data_as(_lib.c_double_p),
            data_bins=self._bins.ctypes.data_as(C.POINTER(C.c_int32)), C=len(self._bins),
            in_dtype=_DT_OF[cfg.in_dtype], max_window=cfg.max_window)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.gf3_ctx_create(C.byref(g), C.byref(h))
        if rc != 0:
            msg = self.lib.gf3_last_error(None).decode()
            raise (ValueError if rc == _lib.GF3_EINVAL else Gf3Error)(msg)
        self._h = h
        self._sync_mode = 0
        self._tls = threading.local()
        self.n_cu = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        self.bytes_per_frame = int(self.lib.gf3_bytes_per_frame(h))
        self.max_window = int(self.lib.gf3_sync_max_window(h))

    def close(self):
        self._tls = threading.local()                     # (drops the calling thread's cached ingest buffers)
        if getattr(self, "_h", None):
            self.lib.gf3_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.gf3_last_error(self._h).decode()
            raise (ValueError if rc == _lib.GF3_EINVAL else Gf3Error)(f"gf3rx error {rc}: {msg}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _samples(self, x):
        """1-D (or any-D contiguous) device tensor of samples in the configured dtype."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if not isinstance(x, torch.Tensor):
            raise TypeError("samples must be a torch tensor or numpy array")
        if x.dtype != self.cfg.in_dtype:
            x = x.to(self.cfg.in_dtype)
        return x.to(self.device).contiguous()

    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ ABI calls
    def chirp_replica(self):
        out = np.empty(self.cfg.chirp_length)
        self._check(self.lib.gf3_chirp_replica(self._h, out.ctypes.data_as(_lib.c_double_p)))
        return out

    def rfft_batch(self, x, offsets, out=None):
        """[n_sym, N/2+1] complex128 spectra of the N samples starting at each offset."""
        x = self._samples(x)
        offsets = torch.as_tensor(offsets, dtype=torch.int64).to(self.device).contiguous()
        n = offsets.numel()
        if out is None:
            out = self._new((n, self.cfg.N // 2 + 1), torch.complex128)
        elif out.dtype != torch.complex128 or out.numel() != n * (self.cfg.N // 2 + 1) or not out.is_contiguous():
            raise ValueError("out must be a contiguous complex128 tensor of n_sym * (N/2+1) elements")
        self._check(self.lib.gf3_rfft_batch(self._h, _ptr(x), x.numel(), _ptr(offsets), n, _ptr(out), self._stream()))
        return out

    def demod_frames(self, x, frame_offsets, want=(), out_bits=None, split=None):
        """Fused a3-a10.  Returns dict with 'bits' (packed uint8
        [F, bytes_per_frame]) plus any of 'eq','Hs','He','slope','Hest','status'.
        split: None -- the library chooses between one packet per workgroup and the two-phase form for long packets,
        few at a time (gf3_demod_frames_ex: pilot sums, estimate, data symbols spread over the chip; the reference's own
        geometry of 3 packets x 220 symbols); False / True force the one or the other."""
        cfg = self.cfg
        x = self._samples(x)
        off = torch.as_tensor(frame_offsets, dtype=torch.int64).to(self.device).contiguous()
        F = off.numel()
        bits = out_bits if out_bits is not None else self._new((F, self.bytes_per_frame), torch.uint8)
        o = {"bits": bits}
        if "eq" in want: o["eq"] = self._new((F * cfg.D, cfg.C), torch.complex128)
        if "Hs" in want: o["Hs"] = self._new((F, cfg.K), torch.complex128)
        if "He" in want: o["He"] = self._new((F, cfg.K), torch.complex128)
        if "slope" in want: o["slope"] = self._new((F,), torch.float64)
        if "Hest" in want: o["Hest"] = self._new((F, cfg.D, cfg.K), torch.complex128)
        if "status" in want: o["status"] = torch.zeros((1,), dtype=torch.int32, device=self.device)
        # the two-phase form needs a workspace (pilot sums, and Hs / He / slope when they are not asked for): allocated exactly
        # when the library would choose that form (at most 2 x CUs packets)
        work = None
        if F and (split or (split is None and self.lib.gf3_demod_split_plan(self._h, F, 0, None, None))):
            work = self._new((int(self.lib.gf3_demod_workspace_bytes(self._h, F)),), torch.uint8)
        self._check(self.lib.gf3_demod_frames_ex(
            self._h, _ptr(x), x.numel(), _ptr(off), F, _ptr(bits), _ptr(o.get("eq")), _ptr(o.get("Hs")),
            _ptr(o.get("He")), _ptr(o.get("slope")), _ptr(o.get("Hest")), _ptr(o.get("status")),
            _ptr(work), 0 if split is None else (2 if split else 1), self._stream()))
        return o

    def demod_plan(self, F, split=None):
        """How demod_frames(F packets, split=...) runs: dict(split: two-phase form or not, Dc: data symbols per workgroup of
        its data stage, chunks: workgroups per packet)."""
        dc, nch = C.c_int32(0), C.c_int32(0)
        two = self.lib.gf3_demod_split_plan(self._h, int(F), 0 if split is None else (2 if split else 1), C.byref(dc), C.byref(nch))
        return dict(split=bool(two), Dc=int(dc.value), chunks=int(nch.value))

    def equalise(self, data, start, end, want=("Hest",)):
        """receiver.equalise on frequency-domain symbols [F,D,K], [F,P,K], [F,P,K]."""
        cfg = self.cfg
        dev = lambda a: torch.as_tensor(a, dtype=torch.complex128).to(self.device).contiguous()
        data, start, end = dev(data), dev(start), dev(end)
        F = data.shape[0]
        if tuple(data.shape) != (F, cfg.D, cfg.K) or tuple(start.shape) != (F, cfg.P, cfg.K) or tuple(end.shape) != (F, cfg.P, cfg.K):
            raise ValueError("equalise: shapes must be [F,D,K], [F,P,K], [F,P,K]")
        o = {"eq_all": self._new((F * cfg.D, cfg.K), torch.complex128),
             "Hs": self._new((F, cfg.K), torch.complex128), "He": self._new((F, cfg.K), torch.complex128),
             "slope": self._new((F,), torch.float64), "bits": self._new((F, self.bytes_per_frame), torch.uint8)}
        if "Hest" in want: o["Hest"] = self._new((F, cfg.D, cfg.K), torch.complex128)
        self._check(self.lib.gf3_equalise(
            self._h, _ptr(data), _ptr(start), _ptr(end), F, _ptr(o["eq_all"]), _ptr(o["Hs"]), _ptr(o["He"]),
            _ptr(o["slope"]), _ptr(o.get("Hest")), _ptr(o["bits"]), self._stream()))
        return o

    def sync_frames(self, x, F, stride, win_lo, win_hi, want_peak=False, out_starts=None, screened=False, work=None):
        """Batched windowed chirp sync: first-pilot sample index per frame (int64, -1 = none).
        out_starts: optional preallocated int64 [F] device tensor to write into.
        screened: evaluate the windows in fp32 with a proven bound first (gf3_sync_frames_ex mode 1) and run the fp64
        kernel only on the windows the bound cannot decide: the same indices.  work: optional preallocated uint8 workspace
        (gf3_sync_frames_workspace_bytes; its first int32 then holds the number of windows that went to fp64)."""
        x = self._samples(x)
        if out_starts is not None:
            if out_starts.dtype != torch.int64 or out_starts.numel() != F or not out_starts.is_contiguous():
                raise ValueError("out_starts must be a contiguous int64 tensor of F elements")
            starts = out_starts
        else:
            starts = self._new((F,), torch.int64)
        peak = self._new((F,), torch.float64) if want_peak else None
        if screened and work is None:
            work = self.sync_frames_workspace(F)
        self._check(self.lib.gf3_sync_frames_ex(self._h, _ptr(x), x.numel(), F, stride, win_lo, win_hi,
                                                _ptr(starts), _ptr(peak), 1 if screened else 0, _ptr(work) if screened else None, self._stream()))
        return (starts, peak) if want_peak else starts

    def sync_frames_workspace(self, F):
        return self._new((int(self.lib.gf3_sync_frames_workspace_bytes(self._h, F)),), torch.uint8)

    def debug_frames_screen(self, x, F, stride, win_lo, win_hi):
        """The fp32 screening pass of the frames sync alone (tests): dict(starts int64 [F] (resolved windows only), y32
        [F, W] float32, err [F] float32 -- the bound on |y32 - exact| --, cls int32 [F] (0 resolved with a detection, 1
        resolved without, 2 unresolved), unresolved: sorted window numbers the fp64 kernel would be run on)."""
        x = self._samples(x)
        W = win_hi - win_lo
        o = dict(starts=torch.full((F,), -7, dtype=torch.int64, device=self.device), y32=self._new((F, W), torch.float32),
                 err=self._new((F,), torch.float32), cls=self._new((F,), torch.int32))
        work = self.sync_frames_workspace(F)
        self._check(self.lib.gf3_debug_frames_screen(self._h, _ptr(x), x.numel(), F, stride, win_lo, win_hi, _ptr(o["starts"]),
                                                     _ptr(o["y32"]), _ptr(o["err"]), _ptr(o["cls"]), _ptr(work), self._stream()))
        n = int(work[:4].view(torch.int32).item())
        o["unresolved"] = torch.sort(work[64: 64 + 4 * n].view(torch.int32)).values
        return o

    def sync_stream(self, x, cap=None, want_corr=False, mode=None, want_info=False):
        """chirp_method on one stream: indices i with zeros[i] True (int64 tensor).
        mode: how the matched filter is evaluated for THIS call (gf3_sync_stream_ex; None = the engine's default set by
        sync_stream_mode): 0 fp32 screening + fp64 decisions from 2^23 samples on, all-fp64 below; 1 always the all-fp64
        overlap-save; 2 screened at any length; 3 as 2 with the general screening kernel.  want_info adds the call's
        diagnostics dict (as sync_stream_info) to the result.  Nothing of a call is stored in the C context, so several
        threads may run this on one engine, each on its own stream."""
        x = self._samples(x).reshape(-1)
        n = x.numel()
        Lc = self.cfg.chirp_length
        cap = cap or max(4, n // Lc + 4)
        peaks = self._new((cap,), torch.int64)
        ws = int(self.lib.gf3_sync_stream_workspace_bytes(self._h, n))
        work = self._new((ws,), torch.uint8)
        corr = self._new((n + Lc - 1,), torch.float64) if want_corr else None
        cnt = C.c_int64(0)
        info = (C.c_int64 * 4)()
        self._check(self.lib.gf3_sync_stream_ex(self._h, _ptr(x), n, _ptr(peaks), cap, C.byref(cnt), _ptr(work),
                                                _ptr(corr), int(self._sync_mode if mode is None else mode), info, self._stream()))
        peaks = peaks[: cnt.value]
        d = dict(path=int(info[0]), cells=int(info[1]), cells_hit=int(info[2]), candidates=int(info[3]))
        self._tls.sync_info = d
        out = (peaks,) + ((corr,) if want_corr else ()) + ((d,) if want_info else ())
        return out[0] if len(out) == 1 else out

    def sync_stream_mode(self, mode):
        """Default `mode` of sync_stream for this Engine object (a Python-side default: the C context is not touched)."""
        if int(mode) not in (0, 1, 2, 3):
            raise ValueError("sync_stream_mode: mode must be 0 (by length), 1 (fp64 only), 2 (always screen) or 3 (always screen, general kernel)")
        self._sync_mode = int(mode)
        self._check(self.lib.gf3_sync_stream_mode(self._h, int(mode)))      # (keeps debug_stream_screen's kernel choice in step)

    def sync_stream_info(self):
        """Of the calling thread's last sync_stream call: dict(path=0 screened | 1 fp64 after a non-selective screen |
        2 fp64, cells = cells of 14 lags re-evaluated in fp64, cells_hit = those holding a candidate, candidates)."""
        return dict(getattr(self._tls, "sync_info", dict(path=2, cells=0, cells_hit=0, candidates=0)))

    def debug_stream_screen(self, x):
        """The fp32 screening pass alone (tests): (P32 [n+Lc-1] float32, block maxima, block error bounds, hop)."""
        x = self._samples(x).reshape(-1)
        n = x.numel()
        plen = n + self.cfg.chirp_length - 1
        p32 = self._new((plen,), torch.float32)
        nb_max = plen // 16 + 2                                  # (hop >= 16: more than enough room)
        blk = self._new((2 * nb_max,), torch.float32)
        hop = C.c_int32(0)
        self._check(self.lib.gf3_debug_stream_screen(self._h, _ptr(x), n, _ptr(p32), _ptr(blk), C.byref(hop), self._stream()))
        nblk = -(-plen // hop.value)
        return p32, blk[:nblk], blk[nblk: 2 * nblk], hop.value

    def tx_frames(self, bits_packed, filler, stride=None, gaps=None, out_dtype=torch.float32):
        """Synthesise chirp-prefixed packets (transmit side of the reference, OFDM.py:196-259).
        bits_packed: uint8 [F, bytes_per_frame] (the format demod_frames writes); filler: complex [K],
        value of every non-data carrier.  Returns [F, stride] samples: row f =
        [gaps[f] zeros | chirp | P known symbols | D data symbols | P known symbols | zeros]."""
        bits = torch.as_tensor(bits_packed, dtype=torch.uint8).to(self.device).contiguous()
        if bits.dim() != 2 or bits.shape[1] != self.bytes_per_frame:
            raise ValueError("bits_packed must be [F, bytes_per_frame]")
        F = bits.shape[0]
        fill = torch.as_tensor(filler, dtype=torch.complex128).to(self.device).contiguous()
        if fill.numel() != self.cfg.K:
            raise ValueError("filler must hold K values (one per carrier)")
        stride = stride or self.cfg.frame_len
        g = None if gaps is None else torch.as_tensor(gaps, dtype=torch.int64).to(self.device).contiguous()
        if g is not None and F and int(g.max()) + self.cfg.frame_len > stride:
            raise ValueError("gap + packet does not fit the row stride")
        out = self._new((F, stride), out_dtype)
        self._check(self.lib.gf3_tx_frames(self._h, _ptr(bits), _ptr(fill), _ptr(g), F, _ptr(out), stride,
                                           _DT_OF[out_dtype], self._stream()))
        return out

    def schmidl_cox(self, x, search_length=None):
        """receiver.schmidlcox_method (OFDM.py:376-387): first arg-max of |P| + N - 1, as a Python int."""
        x = self._samples(x).reshape(-1)
        S = int(5 * self.cfg.fs) if search_length is None else int(search_length)
        out = self._new((1,), torch.int64)
        self._check(self.lib.gf3_schmidl_cox(self._h, _ptr(x), x.numel(), S, _ptr(out), self._stream()))
        return int(out.item())

    def equalise_known_h(self, x, sym_offsets, h):
        """Known-channel zero forcing (Weekend Challenge.ipynb cells 9-17): FFT(rx) / fft(h, N) on the data carriers,
        then demap.  Returns (eq [n_sym, C] complex128, bits [n_sym, C, mu] uint8, idx [n_sym, C] uint8)."""
        x = self._samples(x)
        off = torch.as_tensor(sym_offsets, dtype=torch.int64).to(self.device).contiguous()
        taps = torch.as_tensor(np.asarray(h, dtype=np.float64)).to(self.device).contiguous()
        n = off.numel()
        eq = self._new((n, self.cfg.C), torch.complex128)
        bits = self._new((n, self.cfg.C, self.cfg.mu), torch.uint8)
        idx = self._new((n, self.cfg.C), torch.uint8)
        work = self._new((int(self.lib.gf3_known_h_workspace_bytes(self._h, n)),), torch.uint8)
        self._check(self.lib.gf3_equalise_known_h(self._h, _ptr(x), x.numel(), _ptr(off), n, _ptr(taps), taps.numel(),
                                                  _ptr(eq), _ptr(bits), _ptr(idx), _ptr(work), self._stream()))
        return eq, bits, idx

    def demap_hard(self, sym):
        sym = torch.as_tensor(sym, dtype=torch.complex128).to(self.device).contiguous()
        n = sym.numel()
        bits = self._new(tuple(sym.shape) + (self.cfg.mu,), torch.uint8)
        idx = self._new(tuple(sym.shape), torch.uint8)
        self._check(self.lib.gf3_demap_hard(self._h, _ptr(sym), n, _ptr(bits), _ptr(idx), self._stream()))
        return bits, idx

    def soft_demap(self, sym, noise_var, out=None):
        sym = torch.as_tensor(sym, dtype=torch.complex128).to(self.device).contiguous()
        if out is None:
            llr = self._new(tuple(sym.shape) + (self.cfg.mu,), torch.float32)
        elif out.dtype != torch.float32 or out.numel() != sym.numel() * self.cfg.mu or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 tensor of n * mu elements")
        else:
            llr = out
        self._check(self.lib.gf3_soft_demap(self._h, _ptr(sym), sym.numel(), float(noise_var), _ptr(llr), self._stream()))
        return llr

    # ------------------------------------------------------------------ host ingest (streams from host memory / longer than HBM)
    def receive_host(self, samples, chunk_samples=1 << 24, list_cap=None):
        """chirp sync + demodulation (the arithmetic of receiver.receive, OFDM.py:581-603) of a stream that lives in
        HOST memory, piece by piece: pinned, double-buffered H2D copies on a copy stream run under the kernels of the
        previous piece, and the result is that of the one-shot path -- the reference's rule with the GLOBAL maximum
        (OFDM.py:359), the suppression walk, the except-branch, the dropped last detection -- for any stream length.

        How the global rule survives the cut (gf3_sync_chunk / gf3_sync_decide, include/gf3rx.h): every piece folds its
        lags into a running maximum and keeps the few lags that could still pass 0.4 x the FINAL maximum (which can
        only be larger), with their raw fp64 values.  After each piece the rule is applied provisionally with the
        maximum so far and the packets whose samples are resident are demodulated; at the end it is applied once more
        with the final maximum, and only where that changes the detections (a later piece raised the maximum enough to
        kill an earlier candidate, or un-suppressed one) are packets looked at again -- their samples re-read from the
        host array.  Pieces overlap by Lc + one packet, so no chirp and no packet is cut.

        samples: 1-D numpy array or CPU torch tensor.  A pinned tensor is copied from directly.  Pageable memory of 128 MiB
        and more goes to the runtime in equal pieces of at least 128 MiB (which it pins on the fly: the DMA rate; a copy
        thread makes these blocking copies under the previous piece's kernels); less than that is staged through THREE
        pinned buffers by a host copy per piece that a background thread makes two pieces ahead of the kernels: under piece
        c's kernels and piece c+1's DMA, piece c+2 is being staged -- into the buffer piece c-1 was copied from, which is
        idle by then, so that thread makes no HIP call at all.  (Until round 4 a pageable array could also be registered
        with the driver -- hipHostRegister -- for the duration of the call.  Registering and releasing ordinary process
        memory over and over was followed, in this package's own test runs, by GPU memory faults in unrelated kernels
        later in the process, and the default is as fast now: removed)  chunk_samples: new samples per piece
        (raised to two packets if smaller).  Returns dict(peaks int64 [n_det] (device), bits uint8 [n_det - 1,
        bytes_per_frame] (device, packed), info).  Raises ValueError where the reference fails (fewer than two detections;
        a packet that runs past the end of the stream)."""
        import time
        t_start = time.perf_counter()
        cfg = self.cfg
        if isinstance(samples, torch.Tensor):
            if samples.is_cuda:
                raise ValueError("receive_host takes host memory; use sync_stream / demod_frames for device tensors")
            x = samples.reshape(-1)
            if x.dtype != cfg.in_dtype:
                x = x.to(cfg.in_dtype)
        else:
            a = np.asarray(samples).reshape(-1)
            want = torch.empty(0, dtype=cfg.in_dtype).numpy().dtype
            b = np.ascontiguousarray(a if a.dtype == want else a.astype(want))
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                    # (torch warns when it wraps a non-writable array; it is only read)
                x = torch.from_numpy(b)
        if x.numel() < 3:
            raise ValueError("stream too short")
        try:
            return self._receive_host(x, x.is_pinned(), chunk_samples, list_cap, t_start)
        except BaseException:
            held = getattr(self._tls, "ingest", None) or {}
            for fut in list(held.pop("pending", [])):              # copies still being made on a helper thread
                try:
                    fut.result()
                except Exception:
                    pass
            torch.cuda.synchronize(self.device)                    # nothing of the call is in flight when its buffers are let go
            raise

    def _receive_host(self, x, pinned_in, chunk_samples, list_cap, t_start):
        import time
        cfg = self.cfg
        n = x.numel()
        Lc, L = cfg.chirp_length, cfg.M * cfg.S
        # Pageable memory, a large stream: pieces of at least 128 MiB, copied by the runtime itself.  From that size on a plain
        # copy from pageable memory is pinned by the runtime on the fly and runs at the DMA rate (55 GB/s measured; below
        # it, it is staged at 13-15 GB/s) -- the path every large host-to-device copy of every program takes.  Such a copy
        # blocks its caller, so a copy thread makes it while the calling thread runs the previous piece's kernels.
        direct = False
        if not pinned_in and n * x.element_size() >= DIRECT_PIECE_BYTES + 65536:
            direct = True
            min_piece = -(-(DIRECT_PIECE_BYTES + 65536) // x.element_size())
            k = max(1, min(n // min_piece, -(-n // max(int(chunk_samples), 1))))     # equal pieces, none below the threshold,
            chunk_samples = -(-n // k)                                                # no more of them than were asked for
        pieces, H, carry = host_pieces(n, chunk_samples, Lc, L)  # carry: samples of the previous piece kept in front of a piece
        nchunks = len(pieces)
        plen = n + Lc - 1
        dev = self.device
        main = torch.cuda.current_stream(dev)
        cap_list = int(list_cap or max(4096, 64 * (n // Lc + 2)))
        cap_peaks = n // Lc + 8
        nbuf = carry + min(H, n)
        # Device buffers, workspace, pinned staging, the copy stream and its events are kept between calls (per host
        # thread: two threads may ingest through one Engine at once) and reused while the sizes fit: a receiver that is
        # fed one recording after another does not allocate per call.
        staging = not pinned_in and not direct
        key = (nbuf, cap_list, cap_peaks, bool(staging), bool(direct), cfg.in_dtype, min(3, nchunks))
        res = getattr(self._tls, "ingest", None)
        if res is None or res["key"] != key:
            # (the copy stream is a HIGH-PRIORITY stream: the runtime multiplexes streams of one priority onto a handful of
            #  hardware queues, and a copy stream that lands on the compute stream's queue serialises the upload of piece c+1
            #  behind the kernels of piece c -- 3.2 instead of 2.3 ms per 128 MB piece, seen in the bench process once enough
            #  other streams existed; queues of another priority level are never shared with it)
            res = dict(key=key, copier=torch.cuda.Stream(dev, priority=-1),
                       bufs=[self._new((nbuf,), cfg.in_dtype) for _ in range(min(2, nchunks))],
                       stage=[torch.empty((min(H, n),), dtype=cfg.in_dtype).pin_memory() for _ in range(min(3, nchunks))] if staging else None,
                       idx_all=self._new((cap_list,), torch.int64), val_all=self._new((cap_list, 3), torch.float64),
                       work=self._new((int(self.lib.gf3_sync_chunk_workspace_bytes(self._h, nbuf)),), torch.uint8),
                       peaks_dev=self._new((cap_peaks,), torch.int64),
                       dwork=self._new((int(self.lib.gf3_sync_decide_workspace_bytes(self._h, cap_list)),), torch.uint8),
                       rows=self._new((cap_peaks, self.bytes_per_frame), torch.uint8))
            res["ev_copied"] = [torch.cuda.Event() for _ in res["bufs"]]
            if direct:
                from concurrent.futures import ThreadPoolExecutor
                res["copy_thread"] = ThreadPoolExecutor(1)         # makes the blocking copies, so that the kernels of the previous piece run under them
            if staging:
                from concurrent.futures import ThreadPoolExecutor
                res["pool"] = ThreadPoolExecutor(4)                # the four slices of one staging copy
                res["stager"] = ThreadPoolExecutor(1)              # the staging copy of a piece, off the calling thread
            self._tls.ingest = res
        copier, bufs, stage, ev_copied = res["copier"], res["bufs"], res["stage"], res["ev_copied"]
        idx_all, val_all, work, peaks_dev, dwork, rows = res["idx_all"], res["val_all"], res["work"], res["peaks_dev"], res["dwork"], res["rows"]
        copier.wait_stream(main)                                  # (a previous call's consumers of these buffers are ordered before the new copies)
        run_max = torch.full((2,), float("-inf"), dtype=torch.float64, device=dev)     # [maximum so far, the last piece's own maximum]
        row_of, next_row = {}, 0                                  # zeros-index of a detection -> row of `rows` holding its packet's bits
        segs, overflow = [], []                                   # per piece: (first entry, entries) of the kept list; pieces whose list did not fit
        n_listed = 0
        BIG = (1 << 62)
        info = dict(chunks=nchunks, chunk_samples=H, overlap_samples=carry, pinned_input=bool(pinned_in),
                    source="pinned" if pinned_in else ("pageable, copied by the runtime in large pieces" if direct else "pageable, staged"), h2d_bytes=0,
                    second_look_chunks=0, second_look_packets=0, provisional_detections_dropped=0, full_list_pieces=0)

        def geometry(c):
            q = pieces[c]
            return q["lo"], q["hi"], q["lo"] - q["base"], q["base"], q["g_lo"], q["g_hi"]

        staged = {}                                               # piece -> future of its host-side staging copy
        copies = {}                                               # piece -> future of its blocking copy (large pageable streams)

        def stage_piece(c):
            """pageable source, a small stream: piece c's new samples -> pinned staging buffer c % 3 (a host copy by
            four threads).  Runs on the stager thread, TWO pieces ahead of the kernels -- under piece c - 2's kernels and
            piece c - 1's DMA.  The buffer was last read by the DMA of piece c - 3, which the calling thread has seen
            finish (it synchronised on piece c - 3's kernels, which waited for that DMA) before it submits this: a plain
            host memcpy, no HIP call on this thread."""
            lo_s, hi_s = c * H, min(n, (c + 1) * H)
            src, dst, m = x[lo_s:hi_s], stage[c % 3], hi_s - lo_s
            if m >= (1 << 22):                                     # four host threads: 24 GB/s on the GPU box against 4 GB/s for one
                q = -(-m // 4)
                list(res["pool"].map(lambda k: dst[k * q: min(m, (k + 1) * q)].copy_(src[k * q: min(m, (k + 1) * q)]), range(4)))
            else:
                dst[:m].copy_(src)

        def issue_copy(c):
            """host -> dev of piece c's new samples on the copy stream (after `ev_order` of the main stream)"""
            b = c % 2
            lo_s, hi_s = c * H, min(n, (c + 1) * H)
            src = x[lo_s:hi_s]
            if stage is not None:
                staged.pop(c).result()                            # (staged while the previous piece's kernels ran)
                src = stage[c % 3][: hi_s - lo_s]
            info["h2d_bytes"] += (hi_s - lo_s) * x.element_size()
            if direct:
                # the source is pageable: the copy blocks its caller while the runtime pins and transfers the piece -- so it is
                # made on the copy thread (stream order: the wait for `ev_order` was enqueued by the calling thread before this)
                def job(b=b, src=src, m=hi_s - lo_s):
                    with torch.cuda.stream(copier):
                        bufs[b][carry: carry + m].copy_(src)
                        ev_copied[b].record(copier)
                copies[c] = res["copy_thread"].submit(job)
                return
            with torch.cuda.stream(copier):
                bufs[b][carry: carry + (hi_s - lo_s)].copy_(src, non_blocking=True)
                ev_copied[b].record(copier)

        def sync_piece(buf, n_buf, lag_lo, lag_hi, base, idx_t, val_t, cap):
            """-> (entries listed, or -(entries wanted) - 1 when they do not fit; the piece's own maximum)"""
            cnt, pmax = C.c_int64(0), C.c_double(0.0)
            rc = self.lib.gf3_sync_chunk(self._h, _ptr(buf), n_buf, lag_lo, lag_hi, base, _ptr(run_max), _ptr(idx_t), _ptr(val_t),
                                         cap, C.byref(cnt), C.byref(pmax), _ptr(work), self._stream())
            if rc == _lib.GF3_ERANGE:
                return -int(cnt.value) - 1, pmax.value
            self._check(rc)
            return int(cnt.value), pmax.value

        def decide(idx_t, val_t, k, nz):
            cnt = C.c_int64(0)
            self._check(self.lib.gf3_sync_decide(self._h, _ptr(idx_t), _ptr(val_t), k, _ptr(run_max), nz, _ptr(peaks_dev), cap_peaks,
                                                 C.byref(cnt), _ptr(dwork), self._stream()))
            return peaks_dev[: cnt.value].cpu().numpy()

        info["setup_seconds"] = time.perf_counter() - t_start    # (pinned staging, device buffers, workspace: cached by torch after the first call)
        for fut in list(res.pop("pending", [])):                   # (a previous call that ended in an exception may have left copies running)
            try:
                fut.result()
            except Exception:
                pass
        if stage is not None:
            for c0 in range(min(2, nchunks)):
                staged[c0] = res["stager"].submit(stage_piece, c0)
            res["pending"] = staged.values()
        if direct:
            res["pending"] = copies.values()
        issue_copy(0)
        for c in range(nchunks):
            b = c % 2
            lo_s, hi_s, ce, base, g_lo, g_hi = geometry(c)
            if direct:
                copies.pop(c).result()                             # (the copy thread has recorded ev_copied[b])
            main.wait_event(ev_copied[b])
            if ce:
                # (the previous piece was a full one: its last `ce` new samples sit at the end of its buffer)
                bufs[b][carry - ce: carry].copy_(bufs[1 - b][carry + H - ce: carry + H])
            if c + 1 < nchunks:
                ev_order = torch.cuda.Event()
                ev_order.record(main)                              # the other buffer is free once this point is reached
                copier.wait_event(ev_order)
                issue_copy(c + 1)                                  # its DMA runs under this piece's kernels
            if stage is not None and c + 2 < nchunks:
                # ... and piece c + 2 is staged meanwhile, into the buffer piece c - 1 was copied from: that DMA is
                # known to be finished (this thread synchronised on piece c - 1's kernels, which had waited for it)
                staged[c + 2] = res["stager"].submit(stage_piece, c + 2)
            buf = bufs[b][carry - ce: carry + (hi_s - lo_s)]
            room = cap_list - n_listed                             # (a full list: the piece can only report that it overflows, or keep nothing)
            info["full_list_pieces"] += int(room == 0)
            got, pmax = sync_piece(buf, buf.numel(), g_lo - base, g_hi - base, base, idx_all[n_listed:] if room else None,
                                   val_all[n_listed:] if room else None, room)
            if got < 0:
                # no positive maximum yet (leading silence), or one so small that most lags of this piece stay above 0.4 x
                # it (leading noise): nothing is kept of the piece but its own maximum, which at the end decides whether it
                # has to be looked at again at all
                overflow.append((c, pmax))
                segs.append((n_listed, 0))
            else:
                segs.append((n_listed, got))
                n_listed += got
            # provisional decision with the maximum so far (a piece that kept nothing is simply not represented: whatever
            # that gets wrong is put right at the end), then the packets whose samples are resident
            pk = decide(idx_all, val_all, n_listed, BIG)
            k0 = int(np.searchsorted(pk, base - 2))                # (detections before this buffer were handled, or wait for the end)
            ready = [int(i) for i in pk[k0:] if int(i) + 2 + L <= hi_s and int(i) not in row_of]
            if ready:
                if next_row + len(ready) > rows.shape[0]:           # (provisional detections that are dropped later use rows too)
                    rows = torch.cat([rows, self._new((max(len(ready), rows.shape[0]), self.bytes_per_frame), torch.uint8)])
                st = torch.tensor([i + 2 - base for i in ready], dtype=torch.int64, device=dev)
                self.demod_frames(buf, st, out_bits=rows[next_row: next_row + len(ready)])
                for k, i in enumerate(ready):
                    row_of[i] = next_row + k
                next_row += len(ready)

        info["pieces_seconds"] = time.perf_counter() - t_start - info["setup_seconds"]
        info["listed"] = n_listed                                 # lags kept over the pieces (a superset of what the final rule can accept)

        # ---- the end of the stream: the maximum is final
        def piece_on_device(c):
            lo_s, hi_s, ce, base, g_lo, g_hi = geometry(c)
            buf = bufs[0][: ce + hi_s - lo_s]
            buf.copy_(x[base:hi_s])                                # (second look: a plain synchronous copy)
            info["h2d_bytes"] += buf.numel() * x.element_size()
            return buf, base, g_lo, g_hi

        extra, scratch = {}, None
        M = float(run_max[0].item())
        for c, pmax in overflow:
            if np.isfinite(M) and M > 0.0 and pmax < cfg.thresh * M * (1.0 - 1e-6):
                info["overflow_pieces_below_threshold"] = info.get("overflow_pieces_below_threshold", 0) + 1
                continue                                           # no lag of that piece can pass thresh x (final maximum): nothing to look at
            buf, base, g_lo, g_hi = piece_on_device(c)
            k = g_hi - g_lo
            if scratch is None or scratch[0].numel() < k:          # one scratch pair for every piece looked at again
                scratch = (self._new((k,), torch.int64), self._new((k, 3), torch.float64))
            got, _ = sync_piece(buf, buf.numel(), g_lo - base, g_hi - base, base, scratch[0], scratch[1], k)
            extra[c] = (scratch[0][:got].clone(), scratch[1][:got].clone())   # (what is kept is what was listed, not k x 32 B)
            info["second_look_chunks"] += 1
        if overflow:
            parts_i, parts_v = [], []
            for c, (s0, k) in enumerate(segs):
                if c in extra:
                    parts_i.append(extra[c][0]); parts_v.append(extra[c][1])
                elif k:
                    parts_i.append(idx_all[s0: s0 + k]); parts_v.append(val_all[s0: s0 + k])
            fi = torch.cat(parts_i) if parts_i else idx_all[:0]
            fv = torch.cat(parts_v).contiguous() if parts_v else val_all[:0]
            if fi.numel() > cap_list:
                dwork = self._new((int(self.lib.gf3_sync_decide_workspace_bytes(self._h, fi.numel())),), torch.uint8)
        else:
            fi, fv = idx_all[:n_listed], val_all[:n_listed]
        peaks = decide(fi.contiguous(), fv, fi.numel(), plen - 2)
        if len(peaks) < 2:
            raise ValueError("need at least one array to concatenate")      # np.vstack([]) in get_symbols (OFDM.py:400)
        det = [int(i) for i in peaks[:-1]]                            # the last detection is always dropped (OFDM.py:395)
        missing = [i for i in det if i not in row_of]
        for i in missing:
            if i + 2 + L > n:
                raise ValueError("packet runs past the end of the stream")
        info["provisional_detections_dropped"] = len(set(row_of) - set(int(i) for i in peaks))   # (accepted with an earlier maximum, rejected by the final one)
        for k0 in range(0, len(missing), 64):                         # second look at single packets: samples re-read from the host
            grp = missing[k0: k0 + 64]
            seg = torch.stack([x[i + 2: i + 2 + L] for i in grp]).to(dev)
            info["h2d_bytes"] += seg.numel() * x.element_size()
            if next_row + len(grp) > rows.shape[0]:
                rows = torch.cat([rows, self._new((len(grp), self.bytes_per_frame), torch.uint8)])
            self.demod_frames(seg.reshape(-1), torch.arange(len(grp), device=dev, dtype=torch.int64) * L,
                              out_bits=rows[next_row: next_row + len(grp)])
            for k, i in enumerate(grp):
                row_of[i] = next_row + k
            next_row += len(grp)
            info["second_look_packets"] += len(grp)
        order = torch.tensor([row_of[i] for i in det], dtype=torch.int64, device=dev)
        bits = rows[order]
        torch.cuda.synchronize(dev)
        info["seconds"] = time.perf_counter() - t_start
        info["max"] = M
        return dict(peaks=torch.from_numpy(peaks).to(dev), bits=bits, info=info)

    def release_host_buffers(self):
        """Drop the calling thread's cached ingest resources (device buffers, workspace, pinned staging, copy stream):
        receive_host keeps them between calls so that a receiver fed one recording after another does not allocate."""
        self._tls.ingest = None

    # ------------------------------------------------------------------ PS + decode (OFDM.py:504-505, 541-544)
    def unpack_decode(self, packed, mask_bits=None, to_host=True):
        """[F, bytes_per_frame] packed decisions -> the int64 0/1 array receiver.receive returns (packet -> symbol ->
        carrier -> bit), XORed with tile(mask_bits)[:len] when a whitening mask is given.  to_host: the kernel writes
        straight into pinned host memory (gf3_unpack_bits) and a torch CPU tensor over that memory is returned -- valid
        once the stream has been synchronised; else a device tensor."""
        packed = packed.contiguous()
        F = packed.shape[0]
        n = F * self.cfg.bits_per_frame
        mask = None
        if mask_bits is not None:
            key = bytes(np.asarray(mask_bits, dtype=np.uint8))
            cached = getattr(self, "_mask_cache", None)
            if cached is None or cached[0] != key:
                cached = self._mask_cache = (key, torch.from_numpy(np.frombuffer(key, dtype=np.uint8).copy()).to(self.device))
            mask = cached[1]
        out = torch.empty(n, dtype=torch.int64, pin_memory=True) if to_host else self._new((n,), torch.int64)
        self._check(self.lib.gf3_unpack_bits(self._h, _ptr(packed), F, _ptr(mask), 0 if mask is None else mask.numel(),
                                             C.c_void_p(out.data_ptr()), self._stream()))
        return out

    # ------------------------------------------------------------------ bit helpers (layout only)
    def unpack_bits(self, packed):
        """[F, bytes_per_frame] uint8 -> [F * D*C*mu] uint8 0/1 (np.unpackbits order), on device."""
        sh = torch.arange(7, -1, -1, device=packed.device, dtype=torch.uint8)
        b = (packed.unsqueeze(-1) >> sh) & 1
        return b.reshape(packed.shape[0], -1)[:, : self.cfg.bits_per_frame].reshape(-1)
