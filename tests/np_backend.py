"""A CPU stand-in for the sharded engine backend (TEST INFRASTRUCTURE): numpy restatement of the
per-shard stages K1..K5 with the ownership rules of alaz_amd/csrc/sg_kernels.h, so that the
exchange logic of alaz_amd.sharded.run_window can be exercised with gloo on CPU tensors and
checked against the unsharded oracle."""
from __future__ import annotations

import numpy as np
import torch

from alaz_amd import replay, sharded
from alaz_amd.weights import F_EDGE, F_HID, F_IN

KNOWN, LABEL, OBIP = 0, 1, 2


def _mean_us(s, c):
    return np.where(c > 0, (s.astype(np.float64) / 1000.0) / np.maximum(c, 1), 0.0)


def _std_us(s, q, c):
    m = _mean_us(s, c)
    v = np.where(c > 0, q.astype(np.float64) / np.maximum(c, 1) - m * m, 0.0)
    return np.sqrt(np.maximum(v, 0.0))


class NumpyBackend:
    def __init__(self, *, pod_ip_to_id, svc_ip_to_id, kind, n_labels, weights, layers, rank, world, ncap, max_obip=64):
        self.pod, self.svc, self.kind = pod_ip_to_id, svc_ip_to_id, np.asarray(kind)
        self.nk, self.nl = len(kind), n_labels
        self.W, self.layers, self.rank, self.world, self.ncap, self.max_obip = weights.astype(np.float32), layers, rank, world, ncap, max_obip
        self.device = torch.device("cpu")
        self.stats_flat = torch.zeros(ncap * 14, dtype=torch.int64)
        self.stats_sum = self.stats_flat[: ncap * 12]; self.stats_max = self.stats_flat[ncap * 12:]
        self.capp = ncap
        self.ob_all = torch.zeros((world, max_obip + 1), dtype=torch.int64)
        self.serve = torch.zeros((world, self.capp + 1), dtype=torch.int64)
        self.rows_in = torch.zeros((world, self.capp, F_HID), dtype=torch.float32)
        self.misrouted = 0
        self.edges = {}
        self.alive = {}

    # ---- K1 ----
    def ingest(self, ev):
        for e in ev:
            sp = self.pod.get(int(e["saddr"]))
            if sp is None:
                continue
            frm = (KNOWN, sp, 0)
            d = int(e["daddr"])
            if d in self.svc: to = (KNOWN, self.svc[d], 0)
            elif d in self.pod: to = (KNOWN, self.pod[d], 0)
            elif int(e["host_label"]) and not (int(e["flags"]) & replay.EV_ALIVE): to = (LABEL, int(e["host_label"]) - 1, 0)
            else: to = (OBIP, 0, d)
            if int(e["flags"]) & replay.EV_ALIVE:                    # open connection: creates the edge, counts no request
                if self._owner_ref(frm) != self.rank:
                    self.misrouted += 1
                    continue
                self.edges.setdefault((frm, to), [0, 0, 0, 0, 0])
                self.alive[(frm, to)] = self.alive.get((frm, to), 0) + 1
                continue
            if int(e["flags"]) & replay.EV_REVERSE:
                frm, to = to, frm
            if self._owner_ref(frm) != self.rank:
                self.misrouted += 1
                continue
            a = self.edges.setdefault((frm, to), [0, 0, 0, 0, 0])
            dur = int(e["duration_ns"]); p = int(e["protocol"]); st = int(e["status"])
            err = (st >= 500) if p in (1, 4) else (st == 2 if p in (3, 5, 7) else False)
            a[0] += 1; a[1] += int(err); a[2] += dur; a[3] = max(a[3], dur); a[4] += (dur // 1000) ** 2

    def _owner_ref(self, node):
        t, v, ip = node
        if t == OBIP:
            return int(sharded.owner_of_obip(np.array([ip], dtype=np.uint32), self.world)[0])
        ref = np.array([(t << 30) | v], dtype=np.uint32)
        return int(replay.hash32(ref)[0] % self.world)

    # ---- backend interface of sharded.run_window ----
    def ob_local(self):
        ips = sorted({n[2] for k in self.edges for n in k if n[0] == OBIP})
        buf = torch.zeros(self.max_obip + 1, dtype=torch.int64)
        buf[0] = len(ips); buf[1:1 + len(ips)] = torch.tensor(ips, dtype=torch.int64)
        return buf

    def _dense(self, node):
        t, v, ip = node
        return v if t == KNOWN else (self.nk + v if t == LABEL else self.nk + self.nl + int(np.searchsorted(self.ob, ip)))

    def owner_of_dense(self, v):
        if v < self.nk + self.nl:
            ref = v if v < self.nk else ((LABEL << 30) | (v - self.nk))
            return int(replay.hash32(np.array([ref], dtype=np.uint32))[0] % self.world)
        return int(sharded.owner_of_obip(np.array([self.ob[v - self.nk - self.nl]], dtype=np.uint32), self.world)[0])

    def close_gathered(self):
        g = self.ob_all.numpy()
        self.ob = np.unique(np.concatenate([g[r, 1:1 + int(g[r, 0])] for r in range(self.world)]).astype(np.int64))
        self.N = self.nk + self.nl + len(self.ob)
        rows = sorted((self._dense(f), self._dense(t), a, self.alive.get((f, t), 0)) for (f, t), a in self.edges.items())
        self.alive_csr = np.array([r[3] for r in rows], dtype=np.int64)
        self.frm = np.array([r[0] for r in rows], dtype=np.int64); self.to = np.array([r[1] for r in rows], dtype=np.int64)
        self.acc = np.array([r[2] for r in rows], dtype=np.int64).reshape(-1, 5)
        s = np.zeros((self.ncap, 12), dtype=np.int64); m = np.zeros((self.ncap, 2), dtype=np.int64)
        cnt, err, sm, mx, sq = (self.acc[:, i] if len(self.acc) else np.zeros(0, np.int64) for i in range(5))
        for col_out, col_in, val in ((0, 1, np.ones(len(self.frm), np.int64)), (2, 3, cnt), (4, 5, err), (6, 7, sm), (8, 9, sq), (10, 11, self.alive_csr)):
            np.add.at(s[:, col_out], self.frm, val); np.add.at(s[:, col_in], self.to, val)
        np.maximum.at(m[:, 0], self.frm, mx); np.maximum.at(m[:, 1], self.to, mx)
        self.stats_flat[: self.ncap * 12] = torch.from_numpy(s.reshape(-1))
        self.stats_flat[self.ncap * 12:] = torch.from_numpy(m.reshape(-1))

    def features(self):
        N = self.N
        s = self.stats_flat[: self.ncap * 12].numpy().reshape(self.ncap, 12)[:N].astype(np.float64)
        si = self.stats_flat[: self.ncap * 12].numpy().reshape(self.ncap, 12)[:N]
        m = self.stats_flat[self.ncap * 12:].numpy().reshape(self.ncap, 2)[:N].astype(np.float64)
        self.out_deg = si[:, 0].copy()
        self.out_stats = si[:, [2, 6, 8]].copy()
        kind = np.zeros(N, dtype=np.int64); kind[: self.nk] = self.kind
        x = np.zeros((N, F_IN), dtype=np.float32)
        x[:, 0] = np.log1p(s[:, 0]); x[:, 1] = np.log1p(s[:, 1]); x[:, 2] = np.log1p(s[:, 2]); x[:, 3] = np.log1p(s[:, 3])
        x[:, 4] = np.log1p(_mean_us(si[:, 6], s[:, 2]) / 1000.0); x[:, 5] = np.log1p(_mean_us(si[:, 7], s[:, 3]) / 1000.0)
        x[:, 6] = np.where(s[:, 2] > 0, s[:, 4] / np.maximum(s[:, 2], 1), 0.0); x[:, 7] = np.where(s[:, 3] > 0, s[:, 5] / np.maximum(s[:, 3], 1), 0.0)
        x[:, 8] = np.log1p(m[:, 0] / 1e6); x[:, 9] = np.log1p(m[:, 1] / 1e6)
        x[:, 10] = kind == 1; x[:, 11] = kind == 2; x[:, 12] = kind == 0
        x[:, 13] = np.log1p(_std_us(si[:, 6], si[:, 8], s[:, 2]) / 1000.0); x[:, 14] = np.log1p(_std_us(si[:, 7], si[:, 9], s[:, 3]) / 1000.0)
        x[:, 15] = 1.0
        x[:, 16] = np.log1p(s[:, 10]); x[:, 17] = np.log1p(s[:, 11])
        self.h = [x] + [np.full((N, F_HID), np.nan, dtype=np.float32) for _ in range(self.layers)]   # NaN = "not valid here"

    def halo_requests(self):
        need = sorted({int(v) for v in self.to if self.out_deg[v] > 0 and self.owner_of_dense(int(v)) != self.rank})
        self.req = torch.zeros((self.world, self.capp + 1), dtype=torch.int64)
        for k in range(self.world):
            g = [v for v in need if self.owner_of_dense(v) == k]
            self.req[k, 0] = len(g); self.req[k, 1:1 + len(g)] = torch.tensor(g, dtype=torch.int64)
        return self.req

    def _layer_weights(self, l):
        off = 0
        for k in range(l):
            fi = F_IN if k == 0 else F_HID
            off += 2 * fi * F_HID + F_HID
        fi = F_IN if l == 0 else F_HID
        Ws = self.W[off: off + fi * F_HID].reshape(fi, F_HID); off += fi * F_HID
        Wn = self.W[off: off + fi * F_HID].reshape(fi, F_HID); off += fi * F_HID
        return Ws, Wn, self.W[off: off + F_HID]

    def layer(self, l):
        hin, hout = self.h[l], self.h[l + 1]
        Ws, Wn, b = self._layer_weights(l)
        for v in range(self.N):
            has_out = self.out_deg[v] > 0
            if has_out and self.owner_of_dense(v) != self.rank:
                continue                                   # arrives by halo exchange (or is never needed here)
            nb = self.to[self.frm == v]
            mean = hin[nb].astype(np.float64).mean(axis=0) if len(nb) else np.zeros(hin.shape[1])
            assert not np.isnan(mean).any(), "a neighbour row was neither computed here nor received"
            hout[v] = np.maximum(hin[v].astype(np.float64) @ Ws.astype(np.float64) + mean @ Wn.astype(np.float64) + b, 0.0)

    def pack(self, l):
        out = torch.zeros((self.world, self.capp, F_HID), dtype=torch.float32)
        for r in range(self.world):
            n = int(self.serve[r, 0]); ids = self.serve[r, 1:1 + n].numpy()
            rows = self.h[l][ids]
            assert not np.isnan(rows).any(), "asked for a row this shard does not own"
            out[r, :n] = torch.from_numpy(np.ascontiguousarray(rows))
        return out

    def unpack(self, l):
        for r in range(self.world):
            n = int(self.req[r, 0])
            if n:
                self.h[l][self.req[r, 1:1 + n].numpy()] = self.rows_in[r, :n].numpy()

    def score(self):
        off = sum(2 * (F_IN if k == 0 else F_HID) * F_HID + F_HID for k in range(self.layers))
        w = self.W.astype(np.float64)
        Wu = w[off: off + F_HID * F_HID].reshape(F_HID, F_HID); off += F_HID * F_HID
        Wv = w[off: off + F_HID * F_HID].reshape(F_HID, F_HID); off += F_HID * F_HID
        We = w[off: off + F_EDGE * F_HID].reshape(F_EDGE, F_HID); off += F_EDGE * F_HID
        b1 = w[off: off + F_HID]; off += F_HID
        w2 = w[off: off + F_HID]; off += F_HID
        b2 = w[off]
        h = self.h[self.layers].astype(np.float64)
        cnt, err, sm, mx, sq = (self.acc[:, i] for i in range(5)) if len(self.acc) else (np.zeros(0, np.int64),) * 5
        c = cnt.astype(np.float64)
        m_e = _mean_us(sm, c); s_e = _std_us(sm, sq, c)
        oc, osum, osq = (self.out_stats[self.frm, i] for i in range(3)) if len(self.frm) else (np.zeros(0),) * 3
        mu = _mean_us(osum, oc.astype(np.float64)); sd = _std_us(osum, osq, oc.astype(np.float64))
        z = ((m_e - mu) / np.maximum(sd, 1.0)).astype(np.float32)
        er = np.where(c > 0, err / np.maximum(c, 1), 0.0).astype(np.float32)
        e = np.zeros((len(self.frm), F_EDGE), dtype=np.float32)
        if len(self.frm):
            e[:, 0] = np.log1p(c); e[:, 1] = np.log1p(m_e / 1000.0); e[:, 2] = np.log1p(s_e / 1000.0); e[:, 3] = np.log1p(mx.astype(np.float64) / 1e6)
            e[:, 4] = er; e[:, 5] = np.log1p(err.astype(np.float64)); e[:, 6] = np.clip(z, -8, 8) * 0.125; e[:, 7] = 1.0
        hu, hv = h[self.frm], h[self.to]
        assert not np.isnan(hu).any() and not np.isnan(hv).any(), "an endpoint row is missing on this shard"
        t = np.maximum(hu @ Wu + b1 + hv @ Wv + e.astype(np.float64) @ We, 0.0)
        s = 1.0 / (1.0 + np.exp(-(t @ w2 + b2)))
        self.rows = [(int(self.frm[i]), int(self.to[i]), tuple(int(x) for x in self.acc[i]), float(np.float32(s[i])), float(z[i]), float(er[i]))
                     for i in range(len(self.frm))]
