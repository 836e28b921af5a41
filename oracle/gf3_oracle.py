"""CPU oracle for the GF3 OFDM receive path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This module is a vectorised NumPy restatement of the demodulation chain of the
reference modem (``/root/reference/OFDM.py``).  It exists so that the HIP
kernels can be checked against something that runs anywhere; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``gf3_audio_modem_amd/`` imports it, and the product
path raises if the HIP library is missing rather than falling back to this.

Parity pinning: every stage below is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the
unmodified reference in the build container (``tests/golden/make_golden.py``),
including the reference's own known-answer record (BER 0.023375665289067146,
``Final System Test.ipynb:160``).

Each function cites the reference lines it restates.  Third-party arithmetic
the reference delegates to NumPy/SciPy (``np.fft.fft``, ``np.unwrap``,
``np.angle``, ``np.polyfit``, ``scipy.signal.chirp/convolve``) is restated
explicitly where the HIP side has to re-implement it (``unwrap_rows``,
``ls_slope``, ``chirp_replica``, ``matched_filter``) and the explicit forms are
tested against the library calls.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np

TWO_PI = 2.0 * np.pi


# --------------------------------------------------------------------------
# parameters (restates CamG.__init__, OFDM.py:18-101, generalised per SURVEY §2 row 1)
# --------------------------------------------------------------------------

def qpsk_table():
    """QPSK Gray table in the reference's insertion order (OFDM.py:72-77)."""
    pts = np.array([(1 + 1j) / np.sqrt(2), (1 - 1j) / np.sqrt(2),
                    (-1 - 1j) / np.sqrt(2), (-1 + 1j) / np.sqrt(2)])
    bits = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.int64)
    return pts, bits


def _gray(n):
    return n ^ (n >> 1)


def square_qam_table(mu):
    """Gray-coded square QAM (not in the reference, which is table-generic:
    demap OFDM.py:484-500 works on any mapping_table).  Order = ascending
    integer label, label bits MSB first; first mu/2 bits pick the I level,
    last mu/2 the Q level; unit average energy."""
    assert mu % 2 == 0 and mu >= 2
    h = mu // 2
    L = 1 << h
    lv = np.zeros(L)
    for i in range(L):
        lv[_gray(i)] = 2 * i - (L - 1)          # Gray label -> amplitude level
    scale = np.sqrt(2.0 * (L * L - 1) / 3.0)
    pts, bits = [], []
    for lab in range(1 << mu):
        bi, bq = lab >> h, lab & (L - 1)
        pts.append((lv[bi] + 1j * lv[bq]) / scale)
        bits.append([(lab >> (mu - 1 - b)) & 1 for b in range(mu)])
    return np.array(pts), np.array(bits, dtype=np.int64)


@dataclass
class RxParams:
    N: int = 4096                 # ofdm_symbol_size   OFDM.py:27
    CP: int = 224                 # cp_length          OFDM.py:42
    P: int = 20                   # no_pilots          OFDM.py:51
    D: int = 180                  # packet_length      OFDM.py:50
    lo: int = 1                   # lowest_bin         OFDM.py:43
    hi: int = 2047                # highest_bin (exclusive, np.arange) OFDM.py:44,47
    const_points: np.ndarray = field(default_factory=lambda: qpsk_table()[0])
    const_bits: np.ndarray = field(default_factory=lambda: qpsk_table()[1])
    known_bits: np.ndarray = None  # known_sequence     OFDM.py:99-101
    fs: float = 48000.0           # OFDM.py:24
    f0: float = 0.0               # OFDM.py:62
    f1: float = 8000.0            # OFDM.py:63
    thresh: float = 0.4           # OFDM.py:361
    fit_lo: int = 500             # OFDM.py:462
    fit_hi: int = 1000            # OFDM.py:462

    @property
    def K(self):                  # OFDM.py:28
        return self.N // 2 - 1

    @property
    def S(self):
        return self.N + self.CP

    @property
    def M(self):
        return 2 * self.P + self.D

    @property
    def Lc(self):                 # chirp_length  OFDM.py:64
        return 5 * self.S

    @property
    def mu(self):
        return int(self.const_bits.shape[1])

    @property
    def data_carriers(self):      # OFDM.py:47
        return np.arange(self.lo, self.hi)

    @property
    def C(self):
        return self.hi - self.lo

    @property
    def frame_len(self):
        """samples of one packet including its chirp"""
        return self.Lc + self.M * self.S

    def known_symbols(self):
        """map(known_sequence[:K*mu]) -> K constellation points (OFDM.py:429,196-197)."""
        kb = np.asarray(self.known_bits[: self.K * self.mu]).reshape(self.K, self.mu)
        return map_bits(kb, self)


def load_known_bits(path, count):
    """First `count` ASCII '0'/'1' characters of random_bits.txt (OFDM.py:99-101),
    tiled if the file is shorter (SURVEY §8c re-parameterisation note)."""
    raw = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    raw = raw[(raw == 48) | (raw == 49)] - 48
    if len(raw) < count:
        raw = np.tile(raw, -(-count // len(raw)))
    return raw[:count].astype(np.uint8)


# --------------------------------------------------------------------------
# sync  (OFDM.py:106-109, 356-372, 391-403)
# --------------------------------------------------------------------------

def chirp_replica(p: RxParams):
    """sync_chirp (OFDM.py:106-109): scipy.signal.chirp(method='linear') is
    cos(2*pi*(f0*t + 0.5*beta*t*t)), beta=(f1-f0)/t1, on t=linspace(0,Lc/fs,Lc)
    (endpoint included), then /5."""
    Lc = p.Lc
    t1 = Lc / p.fs
    t = np.linspace(0, t1, Lc)
    beta = (p.f1 - p.f0) / t1
    phase = 2 * np.pi * (p.f0 * t + 0.5 * beta * t * t)
    return np.cos(phase) / 5


def matched_filter(r, p: RxParams):
    """P = convolve(r, chirp[::-1], 'full') (OFDM.py:357-358).  SciPy's
    convolve picks its FFT method at these sizes; restated here as one zero-
    padded real FFT product:  P[m] = sum_k r[m-Lc+1+k] * c[k]."""
    c = chirp_replica(p)
    n = len(r) + len(c) - 1
    nfft = 1 << int(np.ceil(np.log2(n)))
    R = np.fft.rfft(np.asarray(r, dtype=np.float64), nfft)
    Cf = np.fft.rfft(c[::-1], nfft)
    return np.fft.irfft(R * Cf, nfft)[:n]


def matched_filter_chunked(r, p: RxParams, log2_fft=22, workers=None):
    """The same full convolution for streams too long for one transform (BASELINE config 3 at full size:
    321 M samples): overlap-save on the CPU, each block one zero-padded real FFT product of 2**log2_fft points
    (scipy.fft, `workers` threads).  Agrees with matched_filter to rounding (tests/test_oracle_golden.py)."""
    import scipy.fft as sfft
    r = np.asarray(r, dtype=np.float64)
    c = chirp_replica(p)
    Lc, n = len(c), len(r)
    nfft = 1 << log2_fft
    B = nfft - Lc + 1                                    # valid outputs per block
    assert B > 0
    Cf = sfft.rfft(c[::-1], nfft)
    out = np.empty(n + Lc - 1)
    for m0 in range(0, n + Lc - 1, B):
        lo = m0 - (Lc - 1)                               # first sample that reaches output m0
        seg = np.zeros(nfft)
        a, b = max(lo, 0), min(lo + nfft, n)
        if b > a:
            seg[a - lo: b - lo] = r[a:b]
        y = sfft.irfft(sfft.rfft(seg, workers=workers) * Cf, nfft, workers=workers)
        m1 = min(m0 + B, n + Lc - 1)
        out[m0:m1] = y[Lc - 1: Lc - 1 + (m1 - m0)]       # linear convolution values (the first Lc-1 wrap around)
    return out


def matched_filter_direct(r, p: RxParams, m_idx):
    """Literal time-domain value of P at the listed full-convolution indices
    (slow; used to cross-check the FFT forms on a handful of lags)."""
    c = chirp_replica(p)
    r = np.asarray(r, dtype=np.float64)
    Lc = len(c)
    out = np.zeros(len(m_idx))
    for j, m in enumerate(m_idx):
        s = m - Lc + 1
        k0, k1 = max(0, -s), min(Lc, len(r) - s)
        if k1 > k0:
            out[j] = np.dot(r[s + k0: s + k1], c[k0:k1])
    return out


def pick_peaks(Pfull, Lc, n, thresh=0.4):
    """Peak rule of chirp_method (OFDM.py:359-370) on a full correlation P of
    length n+Lc-1.  Returns the bool array 'zeros' of length n+Lc-3.

    * normalise by the signed global max (:359)
    * candidate i  <=>  D[i]*D[i+1] <= 0  and  Pn[i+1] > thresh  (:360-361)
    * sequential suppression: an accepted i clears i+1..i+Lc (:364-368), i.e.
      a candidate survives iff it is more than Lc after the last survivor
    * the except-branch (:369-370): clearing past the end of the array raises
      IndexError, the handler wipes zeros[:i+1]; together with the already
      cleared tail that leaves NO detections at all.  Happens iff an accepted
      i >= len(zeros) - Lc.
    """
    Pn = Pfull / np.amax(Pfull)
    Dd = np.diff(Pn)
    cand = np.flatnonzero(((Dd[:-1] * Dd[1:]) <= 0) & (Pn[1:-1] > thresh))
    zeros = np.zeros(len(Pn) - 2, dtype=bool)
    last = None
    for i in cand:
        if last is not None and i <= last + Lc:
            continue
        if i + Lc >= len(zeros):
            zeros[:] = False
            return zeros
        zeros[i] = True
        last = i
    return zeros


def chirp_method(r, p: RxParams):
    """receiver.chirp_method (OFDM.py:356-372)."""
    return pick_peaks(matched_filter(r, p), p.Lc, len(r), p.thresh)


def frame_starts(zeros):
    """First data sample of every packet (OFDM.py:393-395): detections + 2,
    last (terminating chirp) dropped."""
    idx = np.flatnonzero(zeros) + 2
    return idx[:-1]


def gather_frames(r, starts, p: RxParams):
    """get_symbols (OFDM.py:400-403): [F, 2P+D, N+CP] copies of the stream."""
    r = np.asarray(r)
    if len(starts) == 0:
        raise ValueError("need at least one array to concatenate")  # np.vstack([])
    L = p.M * p.S
    rows = []
    for s in starts:
        seg = r[s: s + L]
        if len(seg) != L:
            raise ValueError("packet runs past the end of the stream")
        rows.append(seg)
    return np.stack(rows).reshape(len(starts), p.M, p.S)


# --------------------------------------------------------------------------
# FFT + pilot split (OFDM.py:407-418, 593)
# --------------------------------------------------------------------------

def demod_fft(frames, p: RxParams):
    """remove_cp (:407-408) + np.fft.fft (:593)."""
    return np.fft.fft(frames[:, :, p.CP:])


def split_pilots(X, p: RxParams):
    """get_data (:412-418): bins 1..K; first P / middle D / last P symbols."""
    car = np.arange(1, p.K + 1)
    return X[:, p.P:-p.P, :][:, :, car], X[:, :p.P, :][:, :, car], X[:, -p.P:, :][:, :, car]


# --------------------------------------------------------------------------
# equaliser (OFDM.py:422-480)
# --------------------------------------------------------------------------

def unwrap_rows(ph):
    """np.unwrap along the last axis, written out."""
    ph = np.asarray(ph, dtype=np.float64)
    dd = np.diff(ph, axis=-1)
    ddmod = np.mod(dd + np.pi, TWO_PI) - np.pi
    ddmod = np.where((ddmod == -np.pi) & (dd > 0), np.pi, ddmod)
    corr = ddmod - dd
    corr = np.where(np.abs(dd) < np.pi, 0.0, corr)
    out = ph.copy()
    out[..., 1:] = ph[..., 1:] + np.cumsum(corr, axis=-1)
    return out


def ls_slope(y):
    """Degree-1 least-squares slope over x=0..L-1 in closed form (what the HIP
    kernel computes); np.polyfit(...,1)[0] is the reference call (:462)."""
    y = np.asarray(y, dtype=np.float64)
    L = y.shape[-1]
    x = np.arange(L, dtype=np.float64)
    xm = x - x.mean()
    return (y * xm).sum(axis=-1) / (xm * xm).sum()


def ls_estimate(start, end, p: RxParams):
    """Hs, He = mean over the pilot axis / known symbols (:443-451)."""
    kn = p.known_symbols()
    Hs = (start.real.mean(axis=1) + 1j * start.imag.mean(axis=1)) / kn
    He = (end.real.mean(axis=1) + 1j * end.imag.mean(axis=1)) / kn
    return Hs, He


def phase_slope(Hs, He, p: RxParams):
    """p_i (:454-462): unwrap each angle separately along carriers, subtract,
    fit a line over the python slice [fit_lo:fit_hi]."""
    pd = np.unwrap(np.angle(He)) - np.unwrap(np.angle(Hs))
    seg = pd[:, p.fit_lo:p.fit_hi]
    if seg.shape[1] < 2:
        raise ValueError("phase-slope fit range holds fewer than 2 carriers")
    x = np.arange(seg.shape[1])
    return np.array([np.polyfit(x, seg[i], 1)[0] for i in range(seg.shape[0])])


def channel_model(Hs, He, slope, p: RxParams):
    """Hest[i,l,n] (:466-475): linear |H| interpolation, phase(Hs)+slope*n*f_l,
    f_l=(l+P/2)/(D+P), n = 0-based carrier index."""
    f = ((np.arange(p.D) + p.P / 2) / (p.D + p.P))[None, :, None]
    n = np.arange(p.K)[None, None, :]
    a0 = np.abs(Hs)[:, None, :]
    a1 = np.abs(He)[:, None, :]
    mag = a0 + (a1 - a0) * f
    ph = np.angle(Hs)[:, None, :] + slope[:, None, None] * n * f
    return mag * np.exp(1j * ph)


def equalise(data, start, end, p: RxParams):
    """receiver.equalise (:422-480) -> (eq[F*D,K], Hs, He, Hest)."""
    Hs, He = ls_estimate(start, end, p)
    slope = phase_slope(Hs, He, p)
    Hest = channel_model(Hs, He, slope, p)
    return (data / Hest).reshape(-1, p.K), Hs, He, Hest, slope


# --------------------------------------------------------------------------
# demap (OFDM.py:484-505) and decode (:541-547)
# --------------------------------------------------------------------------

def demap_hard(sym, p: RxParams):
    """Min-distance decision, first index wins (:487-500).  sym: [..., C]."""
    d = np.abs(sym[..., None] - p.const_points)
    idx = d.argmin(axis=-1)
    return p.const_bits[idx], p.const_points[idx]


def xor_decode(bits, p: RxParams):
    """decode, encoding == 'XOR' (:541-544)."""
    nb = p.C * p.mu
    kb = np.tile(np.asarray(p.known_bits[:nb], dtype=np.int64), -(-len(bits) // nb))[: len(bits)]
    return np.bitwise_xor(bits, kb)


def soft_demap_maxlog(sym, noise_var, p: RxParams):
    """Max-log LLR per bit (NOT in the reference -- parity unpinned; pinned only
    by sign(LLR) == hard bit).  LLR>0 means bit 0."""
    d2 = np.abs(sym[..., None] - p.const_points) ** 2          # [..., M]
    out = np.empty(sym.shape + (p.mu,))
    for b in range(p.mu):
        m0 = d2[..., p.const_bits[:, b] == 0].min(axis=-1)
        m1 = d2[..., p.const_bits[:, b] == 1].min(axis=-1)
        out[..., b] = (m1 - m0) / noise_var
    return out


def zf_known_h(r, sym_offsets, h, p: RxParams):
    """Known-channel zero forcing of the reference's older flow (`Weekend Challenge.ipynb` cells 9-17):
    H = np.fft.fft(h, N); symbols = np.fft.fft(rx_no_cp) / H; data carriers; demap.  PARITY UNPINNED: the
    `equalise(OFDM_demod, H)` it called no longer exists in OFDM.py and its input file is missing, so only the
    formula is restated.  -> (eq [n_sym, C], bits [n_sym, C, mu])"""
    r = np.asarray(r, dtype=np.float64)
    sym = np.stack([r[o: o + p.N] for o in sym_offsets])
    H = np.fft.fft(np.asarray(h, dtype=np.float64), p.N)
    eq = (np.fft.fft(sym) / H)[:, p.data_carriers]
    bits, _ = demap_hard(eq, p)
    return eq, bits


# --------------------------------------------------------------------------
# whole receive (OFDM.py:581-657) on explicit frame starts or via sync
# --------------------------------------------------------------------------

def demod_frames(r, starts, p: RxParams):
    """Stages a3-a10 of SURVEY §8(a) for given first-pilot sample indices."""
    frames = gather_frames(r, starts, p).astype(np.float64)
    X = demod_fft(frames, p)
    data, st, en = split_pilots(X, p)
    eq, Hs, He, Hest, slope = equalise(data, st, en, p)
    eq_d = eq[:, p.data_carriers - 1]                              # :603
    bits, hard = demap_hard(eq_d, p)
    return dict(bits=bits.reshape(-1), eq=eq_d, eq_all=eq, Hs=Hs, He=He,
                Hest=Hest, slope=slope, hard=hard, X=X)


def receive(r, p: RxParams):
    zeros = chirp_method(r, p)
    starts = frame_starts(zeros)
    out = demod_frames(r, starts, p)
    out["zeros"] = zeros
    out["starts"] = starts
    return out


def pack_bits(bits, bits_per_frame):
    """MSB-first byte packing per frame, each frame padded to a whole byte
    (the engine's output format; == np.packbits row-wise)."""
    b = np.asarray(bits, dtype=np.uint8).reshape(-1, bits_per_frame)
    return np.packbits(b, axis=1)


# --------------------------------------------------------------------------
# stream synthesiser: restates transmitter.transmit (OFDM.py:296-343) for a
# payload that exactly fills the packets.  Test-input generator only.
# --------------------------------------------------------------------------

def map_bits(bits2d, p: RxParams):
    """transmitter.map (:196-197) for rows of mu bits."""
    bits2d = np.asarray(bits2d, dtype=np.int64)
    w = 1 << np.arange(p.mu - 1, -1, -1)
    lab_of_point = (p.const_bits * w).sum(axis=1)
    lut = np.zeros(1 << p.mu, dtype=complex)
    lut[lab_of_point] = p.const_points
    return lut[(bits2d * w).sum(axis=-1)]


def build_symbols(payload_syms, fill_syms, p: RxParams):
    """build_OFDM_symbol (:207-217): Hermitian-symmetric N-bin rows."""
    X = np.zeros((payload_syms.shape[0], p.N), dtype=complex)
    dc = p.data_carriers
    unused = np.delete(np.arange(1, p.K + 1), dc - 1)
    X[:, dc] = payload_syms
    X[:, unused] = fill_syms
    X[:, -dc] = np.conj(payload_syms)
    X[:, -unused] = np.conj(fill_syms)
    return X


def add_cp(x, p: RxParams):
    return x if p.CP == 0 else np.hstack([x[:, -p.CP:], x])


def tx_frames(bits, fill_syms, p: RxParams):
    """One row per packet: [chirp | P pilots | D data | P pilots], real, x2
    symbol gain (send_to_stream :242-259)."""
    sym = map_bits(np.asarray(bits).reshape(-1, p.C, p.mu), p)
    td = add_cp(np.fft.ifft(build_symbols(sym, fill_syms, p)), p)
    kn = np.zeros((1, p.N), dtype=complex)
    car = np.arange(1, p.K + 1)
    ks = p.known_symbols()
    kn[0, car] = ks
    kn[0, -car] = np.conj(ks)
    kt = add_cp(np.fft.ifft(kn), p)
    pk = td.reshape(-1, p.D, p.S)
    F = pk.shape[0]
    ktile = np.tile(kt, (F, p.P, 1))
    body = 2 * np.hstack([ktile, pk, ktile]).reshape(F, -1)
    chirp = chirp_replica(p)
    return np.hstack([np.tile(chirp, (F, 1)), body]).real


def tx_stream(bits, fill_syms, p: RxParams, gaps=None, lead=0, tail=2):
    """Serial stream: [lead zeros] + per packet [gap_f zeros | frame] +
    [terminating chirp] + [tail zeros]  (SURVEY §8d synthetic input)."""
    rows = tx_frames(bits, fill_syms, p)
    F = rows.shape[0]
    gaps = np.zeros(F, dtype=int) if gaps is None else np.asarray(gaps)
    parts = [np.zeros(lead)]
    for f in range(F):
        parts += [np.zeros(int(gaps[f])), rows[f]]
    parts += [chirp_replica(p), np.zeros(tail)]
    return np.concatenate(parts)


# --------------------------------------------------------------------------
# batched frame buffers (the engine's gf3_sync_frames + gf3_demod_frames path):
# every row is an independent capture holding one chirp-prefixed packet.
# --------------------------------------------------------------------------

def sync_rows(rows, p: RxParams, win_lo, win_hi):
    """Per row: the reference's matched filter (OFDM.py:357-358) over the row, then
    its peak rule (:359-361) restricted to chirp-start lags [win_lo, win_hi):
    normalise by the window maximum, first local extremum above thresh.
    Returns the first-pilot sample index within each row (-1: none)."""
    out = np.full(len(rows), -1, dtype=np.int64)
    for f, row in enumerate(rows):
        P = matched_filter(row, p)
        seg = P[win_lo + p.Lc - 1: win_hi + p.Lc - 1]
        pn = seg / np.amax(seg)
        d = np.diff(pn)
        cand = np.flatnonzero(((d[:-1] * d[1:]) <= 0) & (pn[1:-1] > p.thresh))
        if len(cand):
            out[f] = win_lo + cand[0] + 1 + p.Lc
    return out


def receive_rows(rows, p: RxParams, win_lo, win_hi):
    rows = np.asarray(rows)
    st = sync_rows(rows, p, win_lo, win_hi)
    stride = rows.shape[1]
    out = demod_frames(rows.reshape(-1), st + np.arange(len(rows)) * stride, p)
    out["starts"] = st
    return out


# --------------------------------------------------------------------------
# Schmidl & Cox timing metric (receiver.schmidlcox_method, OFDM.py:376-387; unused by receive())
# --------------------------------------------------------------------------

def schmidl_cox(r, p: RxParams, search_length=None):
    """P[0] = 0, P[d+1] = P[d] + conj(r[d+L]) r[d+2L] - conj(r[d]) r[d+L] for d < search_length-1, L = K+1
    (:379-385); returns the first index of max |P| plus N-1 (:387).  The running sum is evaluated in the
    reference's order ((P + a) - b) by a cumulative sum over the interleaved terms."""
    r = np.asarray(r, dtype=np.float64)
    L = p.K + 1
    S = int(5 * p.fs) if search_length is None else int(search_length)
    if len(r) < S - 1 + 2 * L:
        raise IndexError("stream shorter than the search length + 2L")
    d = np.arange(S - 1)
    a = r[d + L] * r[d + 2 * L]
    b = r[d] * r[d + L]
    inter = np.empty(2 * (S - 1))
    inter[0::2] = a
    inter[1::2] = -b
    P = np.concatenate([[0.0], np.cumsum(inter)[1::2]])
    return int(np.flatnonzero(np.abs(P) == np.amax(np.abs(P)))[0] + p.N - 1)
