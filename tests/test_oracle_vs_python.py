"""The C oracle (oracle/sg_oracle.c) against an independent pure-Python restatement of the same reference functions
(tests/ref_py.py) on random traces with table churn: pods / services added, updated to new IPs, deleted, IPs that are
both a pod and a service, unknown sources, Host-header and raw-IP outbound destinations, AMQP/Redis direction reversal.
Both were written from the Go source separately; a disagreement means one of them misread it."""
import numpy as np
import pytest

from alaz_amd import replay
from tests.ref_py import Aggregator, int_to_ipv4

CLOCK = (1_000_000_000, 1_700_000_000_000_000_000)


@pytest.mark.parametrize("seed", [3, 11, 29])
def test_edge_identity_rows_equal_an_independent_python_restatement(oracle_lib, seed):
    rng = np.random.default_rng(seed)
    n_ip = 60
    ips = [0x0A000001 + i for i in range(n_ip)] + [0xAC100001 + i for i in range(20)]
    labels = [f"host{k}.example.com" for k in range(8)]
    o = oracle_lib.Oracle(*CLOCK, log_limit=1_000_000)
    py = Aggregator(*CLOCK)
    total_ev = 0
    for rnd in range(30):
        # ---- table churn (persist.go) ----
        for _ in range(int(rng.integers(3, 25))):
            kind = "pod" if rng.random() < 0.6 else "svc"
            et = ["ADD", "UPDATE", "DELETE"][int(rng.choice(3, p=[0.55, 0.25, 0.20]))]
            ip = int(ips[int(rng.integers(0, len(ips)))])
            uid = f"{kind}-{int(rng.integers(0, 40))}"
            ip_s = "" if (kind == "pod" and rng.random() < 0.05) else int_to_ipv4(ip)
            (o.pod if kind == "pod" else o.svc)(et, uid, ip_s)
            (py.process_pod if kind == "pod" else py.process_svc)(et, uid, ip_s)
        # ---- a batch of packed events ----
        n = int(rng.integers(50, 400))
        ev = np.zeros(n, dtype=replay.EVENT_DTYPE)
        pick = lambda: np.asarray(ips, dtype=np.uint32)[rng.integers(0, len(ips), n)]
        ev["saddr"] = np.where(rng.random(n) < 0.9, pick(), 0xC0A80000 + rng.integers(0, 50, n)).astype(np.uint32)
        ev["daddr"] = np.where(rng.random(n) < 0.8, pick(), 0x08080000 + rng.integers(0, 30, n)).astype(np.uint32)
        ev["host_label"] = np.where(rng.random(n) < 0.5, rng.integers(1, len(labels) + 1, n), 0)
        ev["protocol"] = rng.choice([1, 2, 3, 4, 5, 6, 7, 8], n)
        ev["status"] = rng.choice([200, 404, 503, 1, 2], n)
        fl = np.where(rng.random(n) < 0.3, replay.EV_TLS, 0)
        rev = ((ev["protocol"] == 2) | (ev["protocol"] == 5)) & (rng.random(n) < 0.5)
        ev["flags"] = (fl | np.where(rev, replay.EV_REVERSE, 0)).astype(np.uint8)
        ev["duration_ns"] = rng.integers(1, 10**10, n)
        ev["write_time_ns"] = CLOCK[0] + rng.integers(-10**9, 10**12, n)          # also before FirstKernelTime: u64 wrap-around
        o.packed(ev, labels)
        for e in ev:
            py.l7(int(e["saddr"]), int(e["daddr"]), labels[int(e["host_label"]) - 1] if e["host_label"] else "", int(e["status"]),
                  int(e["protocol"]), bool(e["flags"] & replay.EV_TLS), bool(e["flags"] & replay.EV_REVERSE), int(e["duration_ns"]),
                  int(e["write_time_ns"]) & ((1 << 64) - 1))
        total_ev += n
    got = o.reqinfos()
    assert len(got) == len(py.rows) > 1000 and o.dropped_src == py.dropped > 0
    for a, b in zip(got, py.rows):
        # ReqInfo slots (backend.go:824-839): 0 StartTime, 1 Latency, 3 FromType, 4 FromUID, 7 ToType, 8 ToUID, 10 Protocol, 11 StatusCode, 15 Tls
        assert (a[0], a[1], a[3], a[4], a[7], a[8], a[10], a[11], a[15]) == b
    kinds = {(r[2], r[4]) for r in py.rows}
    assert {("pod", "service"), ("pod", "pod"), ("pod", "outbound"), ("service", "pod"), ("outbound", "pod")} <= kinds


def test_window_sequence_ledgers_equal_the_python_restatement(oracle_lib):
    """The per-window edge ledger across a SEQUENCE of windows (the warm-window semantics get a second definition
    too): six windows over one set of tables — the same requests again, a subset, requests on other edges, an empty window, the first set
    again — each closed by the oracle; its rows must be exactly the aggregation of the requests of THAT window as tests/ref_py.py
    restates them (an edge no request touched in a window is absent from it, whatever earlier windows held), counts / errors / sums /
    maxima / sums of squares equal."""
    from alaz_amd import weights
    from tests.ref_py import window_ledger
    rng = np.random.default_rng(77)
    ips = [0x0A000001 + i for i in range(40)] + [0xAC100001 + i for i in range(12)]
    labels = [f"host{k}.example.com" for k in range(6)]
    o = oracle_lib.Oracle(*CLOCK)
    py = Aggregator(*CLOCK)
    for i in range(40):
        o.pod("ADD", f"pod-{i}", int_to_ipv4(ips[i])); py.process_pod("ADD", f"pod-{i}", int_to_ipv4(ips[i]))
    for j in range(12):
        o.svc("ADD", f"svc-{j}", int_to_ipv4(ips[40 + j])); py.process_svc("ADD", f"svc-{j}", int_to_ipv4(ips[40 + j]))

    def batch(n, seed):
        r = np.random.default_rng(seed)
        ev = np.zeros(n, dtype=replay.EVENT_DTYPE)
        ev["saddr"] = np.asarray(ips[:40], dtype=np.uint32)[r.integers(0, 40, n)]
        ev["daddr"] = np.where(r.random(n) < 0.85, np.asarray(ips, dtype=np.uint32)[r.integers(0, len(ips), n)], 0x08080000 + r.integers(0, 5, n)).astype(np.uint32)
        ev["host_label"] = np.where(r.random(n) < 0.7, r.integers(1, len(labels) + 1, n), 0)
        ev["protocol"] = r.choice([1, 3, 4, 6], n)
        ev["status"] = np.where(np.isin(ev["protocol"], [1, 4]), r.choice([200, 404, 500, 503], n), r.choice([1, 2], n))
        ev["flags"] = np.where(r.random(n) < 0.2, replay.EV_TLS, 0).astype(np.uint8)
        ev["duration_ns"] = r.integers(1_000, 3 * 10**9, n)
        ev["write_time_ns"] = CLOCK[0] + r.integers(0, 10**9, n)
        return ev
    A, B = batch(3000, 1), batch(3000, 2)
    W = weights.make_weights(1)
    for k, ev in enumerate((A, A, A[::5], B, A[:0], A)):
        o.packed(ev, labels); o.window_close(W, 1)
        py.rows.clear()
        for e in ev:
            py.l7(int(e["saddr"]), int(e["daddr"]), labels[int(e["host_label"]) - 1] if e["host_label"] else "", int(e["status"]),
                  int(e["protocol"]), bool(e["flags"] & replay.EV_TLS), False, int(e["duration_ns"]), int(e["write_time_ns"]))
        want = window_ledger(py.rows)
        got = {kk: v[:5] for kk, v in o.edge_dict().items()}
        assert set(got) == set(want), (k, len(got), len(want))
        assert got == want, k
    assert rng is not None
