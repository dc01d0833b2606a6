"""Shared test harness: drives the oracle and the engine with the same trace and compares them.

The engine speaks node ids; the reference speaks (Type, UID) strings.  HostShim plays the part of
the Go GraphDS shim: it interns UIDs to ids in arrival order and keeps the label
table, so engine rows can be mapped back to the reference's identities."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from alaz_amd import engine as eng
from alaz_amd.replay import ip_str

CLOCK = (1_000_000_000, 1_700_000_000_000_000_000)   # FirstKernelTime, FirstUserspaceTime


class HostShim:
    def __init__(self):
        self.ids: Dict[str, int] = {}
        self.uid_of: List[str] = []
        self.kind: List[str] = []

    def intern(self, uid: str, kind: str) -> int:
        if uid not in self.ids:
            self.ids[uid] = len(self.uid_of); self.uid_of.append(uid); self.kind.append(kind)
        else:
            self.kind[self.ids[uid]] = kind
        return self.ids[uid]

    def apply(self, g: "eng.ServiceGraph", ops: Sequence[Tuple[str, str, str, str]]):
        """ops: (kind, event_type, uid, ip) — aggregator/persist.go processPod / processSvc."""
        for kind, et, uid, ip in ops:
            if kind == "pod" and ip == "":
                continue                      # persist.go:37-40
            nid = self.intern(uid, "pod" if kind == "pod" else "service")
            ipn = eng.ip_u32(ip) if ip else 0
            if kind == "pod":
                g.upsert_pod(ipn, nid) if et in ("ADD", "UPDATE") else g.delete_pod(ipn)
            else:
                g.upsert_service(ipn, nid) if et in ("ADD", "UPDATE") else g.delete_service(ipn)

    def ref_identity(self, ref: int, labels: Sequence[str], obips: np.ndarray) -> Tuple[str, str]:
        t, v = ref >> 30, ref & 0x3FFFFFFF
        if t == eng.REF_KNOWN:
            return self.kind[v], self.uid_of[v]
        if t == eng.REF_LABEL:
            return "outbound", labels[v]
        return "outbound", ip_str(int(obips[v]))


def engine_edge_dict(rows: np.ndarray, shim: HostShim, labels: Sequence[str], obips: np.ndarray):
    d = {}
    for r in rows:
        ft, fu = shim.ref_identity(int(r["from_ref"]), labels, obips)
        tt, tu = shim.ref_identity(int(r["to_ref"]), labels, obips)
        d[(ft, fu, tt, tu)] = (int(r["count"]), int(r["err_count"]), int(r["sum_ns"]), int(r["max_ns"]), int(r["sumsq_us"]),
                               float(r["score"]), float(r["lat_z"]), float(r["err_ratio"]), int(r["alive"]), int(r["p50_us"]), int(r["p99_us"]))
    return d


def compare_edge_dicts(got: dict, want: dict, score_tol: float = 1e-5, percentiles: bool = False):
    """Identities + integer accumulators bit-exact; fp32 outputs within tolerance (north_star:
    1e-5 abs on scores; lat_z 1e-5 relative with an absolute floor; err_ratio is one correctly
    rounded division of exact integers, so it is compared exactly)."""
    assert set(got) == set(want), f"edge identity sets differ: only-engine={list(set(got) - set(want))[:3]} only-oracle={list(set(want) - set(got))[:3]}"
    worst = 0.0
    for k, w in want.items():
        g = got[k]
        assert g[:5] == w[:5], f"integer accumulators differ on {k}: {g[:5]} vs {w[:5]}"
        assert abs(g[5] - w[5]) <= score_tol, f"score differs on {k}: {g[5]} vs {w[5]}"
        assert abs(g[6] - w[6]) <= 1e-5 * max(1.0, abs(w[6])), f"lat_z differs on {k}: {g[6]} vs {w[6]}"
        assert g[7] == w[7], f"err_ratio differs on {k}: {g[7]} vs {w[7]}"
        if len(g) > 8 and len(w) > 8:
            assert g[8] == w[8], f"alive count differs on {k}: {g[8]} vs {w[8]}"
        if percentiles:                                    # f-3: only engines created with edge_histogram report them (the oracle always does)
            assert g[9:11] == w[9:11], f"p50/p99 differ on {k}: {g[9:11]} vs {w[9:11]}"
        worst = max(worst, abs(g[5] - w[5]))
    return worst
