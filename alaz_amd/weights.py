"""Deterministic weight blob for the scoring model.

The model is not trained anywhere (the reference has no scoring code); weights are generated:
U(-1/sqrt(fan_in), +1/sqrt(fan_in)) from splitmix64, one stream per tensor, 24-bit mantissas so
every value is exact in fp32.  The same blob is handed to the engine (sg_load_weights) and, in
tests, to the oracle.

Layout (fp32, in this order):
  for l in 0..L-1:  Ws_l [F_in(l)][64]   Wn_l [F_in(l)][64]   b_l [64]      F_in(0)=32, else 64
  head:             Wu [64][64]  Wv [64][64]  We [8][64]  b1 [64]  w2 [64]  b2 [1]
Row-major, input index k major: W[k][j] is the weight from input k to output j.
"""
from __future__ import annotations

import numpy as np

from .replay import splitmix64

F_IN, F_HID, F_EDGE = 32, 64, 8


def layer_in(l: int) -> int:
    return F_IN if l == 0 else F_HID


def weights_count(layers: int) -> int:
    n = sum(2 * layer_in(l) * F_HID + F_HID for l in range(layers))
    return n + 2 * F_HID * F_HID + F_EDGE * F_HID + F_HID + F_HID + 1


def _tensor(seed: int, stream: int, n: int, fan_in: int) -> np.ndarray:
    u = (splitmix64(seed, n, stream) >> np.uint64(40)).astype(np.float64) / float(1 << 24)   # [0,1)
    return ((u * 2.0 - 1.0) / np.sqrt(float(fan_in))).astype(np.float32)


def make_weights(layers: int, seed: int = 0x5EED_0001) -> np.ndarray:
    parts = []
    s = 0
    for l in range(layers):
        fi = layer_in(l)
        parts.append(_tensor(seed, s, fi * F_HID, 2 * fi)); s += 1      # Ws
        parts.append(_tensor(seed, s, fi * F_HID, 2 * fi)); s += 1      # Wn
        parts.append(_tensor(seed, s, F_HID, 2 * fi)); s += 1           # b
    parts.append(_tensor(seed, 100, F_HID * F_HID, 2 * F_HID + F_EDGE))  # Wu
    parts.append(_tensor(seed, 101, F_HID * F_HID, 2 * F_HID + F_EDGE))  # Wv
    parts.append(_tensor(seed, 102, F_EDGE * F_HID, 2 * F_HID + F_EDGE))  # We
    parts.append(_tensor(seed, 103, F_HID, 2 * F_HID + F_EDGE))          # b1
    parts.append(_tensor(seed, 104, F_HID, F_HID))                       # w2
    parts.append(_tensor(seed, 105, 1, F_HID))                           # b2
    w = np.concatenate(parts).astype(np.float32)
    assert len(w) == weights_count(layers)
    return np.ascontiguousarray(w)
