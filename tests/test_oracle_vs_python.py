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
