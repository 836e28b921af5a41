"""Shared helpers for the tests: fixture loading -> oracle parameter blocks."""
import os

import numpy as np

from oracle import gf3_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def params_of(g, **over):
    kw = dict(N=int(g["N"]), CP=int(g["CP"]), P=int(g["P"]), D=int(g["D"]),
              lo=int(g["lo"]), hi=int(g["hi"]),
              const_points=g["const_points"], const_bits=g["const_bits"].astype(np.int64),
              known_bits=g["known_bits"].astype(np.uint8))
    kw.update(over)
    return orc.RxParams(**kw)


def unpack(g, key="bits", nkey="n_bits"):
    return np.unpackbits(g[key])[: int(g[nkey])]


def modeA2_params(known_bits):
    """The reference defaults: receiver(mode='A2', ...) (OFDM.py:18-51)."""
    pts, bt = orc.qpsk_table()
    return orc.RxParams(N=4096, CP=224, P=20, D=180, lo=100, hi=1500,
                        const_points=pts, const_bits=bt, known_bits=known_bits)


LOOPBACKS = ["g1_n1024_qpsk", "g2_n4096_qpsk", "g3_n4096_16qam_gr5",
             "g7_n4096_qpsk_drift", "g8_n4096_qpsk_gr5_drift"]
