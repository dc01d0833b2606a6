"""Independent numpy restatement of the scoring model — a cross-check of
Part 2 of sg_oracle.c.  TEST INFRASTRUCTURE ONLY.  Vectorised, fp64 statistics, fp32 network in
plain matrix form (no pinned reduction order): agrees with the C oracle to ~1e-6, not bitwise."""
from __future__ import annotations

import numpy as np

F_IN, F_HID, F_EDGE = 32, 64, 8


def _mean_us(s, c):
    return np.where(c > 0, (s.astype(np.float64) / 1000.0) / np.maximum(c, 1), 0.0)


def _std_us(s, q, c):
    m = _mean_us(s, c)
    v = np.where(c > 0, q.astype(np.float64) / np.maximum(c, 1) - m * m, 0.0)
    return np.sqrt(np.maximum(v, 0.0))


def score(n_nodes: int, kind: np.ndarray, frm: np.ndarray, to: np.ndarray, count, err, sum_ns, max_ns, sumsq_us,
          weights: np.ndarray, layers: int):
    """kind[v] in {0 outbound, 1 pod, 2 service}; edges given as dense endpoints + integer accumulators.
    Returns (score, lat_z, err_ratio, x0, h_last)."""
    N = n_nodes
    f8 = np.float64
    def nsum(idx, val): return np.bincount(idx, weights=val.astype(f8), minlength=N)
    ones = np.ones(len(frm))
    out_deg, in_deg = nsum(frm, ones), nsum(to, ones)
    out_cnt, in_cnt = nsum(frm, count), nsum(to, count)
    out_err, in_err = nsum(frm, err), nsum(to, err)
    out_sum, in_sum = nsum(frm, sum_ns), nsum(to, sum_ns)
    out_ssq, in_ssq = nsum(frm, sumsq_us), nsum(to, sumsq_us)
    out_max = np.zeros(N); np.maximum.at(out_max, frm, max_ns.astype(f8))
    in_max = np.zeros(N); np.maximum.at(in_max, to, max_ns.astype(f8))
    x = np.zeros((N, F_IN), dtype=np.float32)
    x[:, 0] = np.log1p(out_deg); x[:, 1] = np.log1p(in_deg); x[:, 2] = np.log1p(out_cnt); x[:, 3] = np.log1p(in_cnt)
    x[:, 4] = np.log1p(_mean_us(out_sum, out_cnt) / 1000.0); x[:, 5] = np.log1p(_mean_us(in_sum, in_cnt) / 1000.0)
    x[:, 6] = np.where(out_cnt > 0, out_err / np.maximum(out_cnt, 1), 0.0); x[:, 7] = np.where(in_cnt > 0, in_err / np.maximum(in_cnt, 1), 0.0)
    x[:, 8] = np.log1p(out_max / 1e6); x[:, 9] = np.log1p(in_max / 1e6)
    x[:, 10] = kind == 1; x[:, 11] = kind == 2; x[:, 12] = kind == 0
    x[:, 13] = np.log1p(_std_us(out_sum, out_ssq, out_cnt) / 1000.0); x[:, 14] = np.log1p(_std_us(in_sum, in_ssq, in_cnt) / 1000.0)
    x[:, 15] = 1.0

    w = weights.astype(np.float32); off = 0
    h = x
    for l in range(layers):
        fi = F_IN if l == 0 else F_HID
        Ws = w[off:off + fi * F_HID].reshape(fi, F_HID); off += fi * F_HID
        Wn = w[off:off + fi * F_HID].reshape(fi, F_HID); off += fi * F_HID
        b = w[off:off + F_HID]; off += F_HID
        agg = np.zeros((N, fi), dtype=np.float64)
        np.add.at(agg, frm, h[to].astype(np.float64))
        mean = np.where(out_deg[:, None] > 0, agg / np.maximum(out_deg[:, None], 1), 0.0).astype(np.float32)
        h = np.maximum(h.astype(np.float64) @ Ws.astype(np.float64) + mean.astype(np.float64) @ Wn.astype(np.float64) + b, 0.0).astype(np.float32)
    Wu = w[off:off + F_HID * F_HID].reshape(F_HID, F_HID); off += F_HID * F_HID
    Wv = w[off:off + F_HID * F_HID].reshape(F_HID, F_HID); off += F_HID * F_HID
    We = w[off:off + F_EDGE * F_HID].reshape(F_EDGE, F_HID); off += F_EDGE * F_HID
    b1 = w[off:off + F_HID]; off += F_HID
    w2 = w[off:off + F_HID]; off += F_HID
    b2 = w[off]
    P = h.astype(np.float64) @ Wu.astype(np.float64) + b1
    Q = h.astype(np.float64) @ Wv.astype(np.float64)
    c = count.astype(np.float64)
    m_e = _mean_us(sum_ns, c); s_e = _std_us(sum_ns, sumsq_us, c)
    mu = _mean_us(out_sum, out_cnt)[frm]; sd = _std_us(out_sum, out_ssq, out_cnt)[frm]
    z = ((m_e - mu) / np.maximum(sd, 1.0)).astype(np.float32)
    er = np.where(c > 0, err / np.maximum(c, 1), 0.0).astype(np.float32)
    e = np.zeros((len(frm), F_EDGE), dtype=np.float32)
    e[:, 0] = np.log1p(c); e[:, 1] = np.log1p(m_e / 1000.0); e[:, 2] = np.log1p(s_e / 1000.0); e[:, 3] = np.log1p(max_ns.astype(f8) / 1e6)
    e[:, 4] = er; e[:, 5] = np.log1p(err.astype(f8)); e[:, 6] = np.clip(z, -8, 8) * 0.125; e[:, 7] = 1.0
    t = np.maximum(P[frm] + Q[to] + e.astype(np.float64) @ We.astype(np.float64), 0.0)
    logit = t @ w2.astype(np.float64) + b2
    s = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    return s, z, er, x, h
