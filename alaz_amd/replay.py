"""Synthetic eBPF event replay: seeded topologies and L7 event streams.

The shape follows the reference's own simulator (main_benchmark_test.go:84-150, 311-343, 561-633;
testconfig/config1.json): fake pods + services announced through k8s ADD events, a fixed set of
edges, and per-edge HTTP request events.  Everything is generated with a counter-based splitmix64
so that a (config, seed) pair names one exact trace on every machine.

Outputs are plain numpy: the packed 32-byte ``sg_event`` records of include/servicegraph.h, the
k8s table operations, the Host-header label table and (for small traces) the full 1096-byte
``struct l7_event`` wire records of ebpf/c/l7.c:19-47 that the reference's perf reader sees.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

# include/servicegraph.h: sg_event
EVENT_DTYPE = np.dtype([
    ("saddr", "<u4"), ("daddr", "<u4"), ("host_label", "<u4"), ("status", "<u2"),
    ("protocol", "u1"), ("flags", "u1"), ("duration_ns", "<u8"), ("write_time_ns", "<u8"),
])
assert EVENT_DTYPE.itemsize == 32

# include/servicegraph.h: sg_edge_out
EDGE_OUT_DTYPE = np.dtype([
    ("sum_ns", "<u8"), ("max_ns", "<u8"), ("sumsq_us", "<u8"), ("from_ref", "<u4"), ("to_ref", "<u4"),
    ("count", "<u4"), ("err_count", "<u4"), ("score", "<f4"), ("lat_z", "<f4"), ("err_ratio", "<f4"),
    ("alive", "<u4"), ("p50_us", "<u4"), ("p99_us", "<u4"),
])
assert EDGE_OUT_DTYPE.itemsize == 64
HIST_BINS = 16


def hist_bin(dur_ns: np.ndarray) -> np.ndarray:
    """include/servicegraph.h: latency histogram bin of a duration in ns (one bin per octave from 2^17 ns)."""
    d = np.maximum(np.asarray(dur_ns, dtype=np.uint64), np.uint64(1))
    lg = np.floor(np.log2(d.astype(np.float64))).astype(np.int64)
    lg = np.where((np.uint64(1) << lg.astype(np.uint64)) > d, lg - 1, lg)          # float rounding at exact powers of two
    lg = np.where((np.uint64(2) << lg.astype(np.uint64)) <= d, lg + 1, lg)
    return np.clip(lg - 16, 0, HIST_BINS - 1).astype(np.int64)

PROTO_HTTP, PROTO_AMQP, PROTO_POSTGRES, PROTO_HTTP2, PROTO_REDIS, PROTO_KAFKA, PROTO_MYSQL, PROTO_MONGO = 1, 2, 3, 4, 5, 6, 7, 8
EV_TLS, EV_REVERSE, EV_CONSUME, EV_ALIVE = 1, 2, 4, 8
L7_WIRE_SIZE = 1096

POD_IP_BASE = 0x0A000000 + 1      # 10.0.0.1 + i
SVC_IP_BASE = 0xAC100000 + 1      # 172.16.0.1 + j
UNKNOWN_SRC_BASE = 0xC0A80000 + 1  # 192.168.0.1 + k  (never registered: events must be dropped)
EXTERNAL_IP_BASE = 0x08080000 + 1  # 8.8.0.1 + k     (outbound destinations)

#: BASELINE.json configs restated (SURVEY.md §8d "Sizes").  layers = L.
CONFIGS = {
    1: dict(pods=50, edges=200, events=10_000, layers=1),
    2: dict(pods=1_000, edges=50_000, events=1_000_000, layers=1),
    3: dict(pods=10_000, edges=1_000_000, events=10_000_000, layers=2),
    4: dict(pods=10_000, edges=1_000_000, events=10_000_000, layers=2),
    5: dict(pods=100_000, edges=20_000_000, events=5_000_000, layers=2),
}
SEED_BASE = 0xA1A2_0000


# ------------------------------------------------------------------------------------------------
# splitmix64, counter based
# ------------------------------------------------------------------------------------------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n outputs of splitmix64 started at ``seed`` (stream selects an independent sequence)."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed + 0x632BE59BD9B4E019 * (stream + 1)) & 0xFFFFFFFFFFFFFFFF)
        z = base + _GOLDEN * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, n: int, stream: int) -> np.ndarray:
    """float64 uniforms in (0,1): 53 random bits, never exactly 0."""
    return ((splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def hash32(x: np.ndarray) -> np.ndarray:
    """murmur3 fmix32 — sg_hash32 of include/servicegraph.h, vectorised."""
    h = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        h ^= h >> np.uint32(16); h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13); h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return h


def _zipf_cdf(n: int, s: float) -> np.ndarray:
    w = np.arange(1, n + 1, dtype=np.float64) ** (-s)
    c = np.cumsum(w)
    return c / c[-1]


def ip_str(ip: int) -> str:
    return f"{(ip >> 24) & 255}.{(ip >> 16) & 255}.{(ip >> 8) & 255}.{ip & 255}"


# ------------------------------------------------------------------------------------------------
# topology
# ------------------------------------------------------------------------------------------------
@dataclass
class Topology:
    n_pods: int
    n_svcs: int
    pod_ips: np.ndarray           # u32 [P]
    svc_ips: np.ndarray           # u32 [S]
    edge_src: np.ndarray          # pod index [E]
    edge_dst: np.ndarray          # node index [E]: < P pod, >= P service
    seed: int
    labels: List[str] = field(default_factory=list)

    @property
    def n_nodes(self) -> int:
        return self.n_pods + self.n_svcs

    def node_ip(self, idx: np.ndarray) -> np.ndarray:
        idx = np.asarray(idx)
        return np.where(idx < self.n_pods, self.pod_ips[np.minimum(idx, self.n_pods - 1)],
                        self.svc_ips[np.clip(idx - self.n_pods, 0, self.n_svcs - 1)]).astype(np.uint32)

    def pod_uid(self, i: int) -> str:
        return f"{i:08x}-0000-4000-8000-00aa{i:08x}"

    def svc_uid(self, j: int) -> str:
        return f"{j:08x}-0000-4000-9000-00bb{j:08x}"

    def node_uid(self, idx: int) -> str:
        return self.pod_uid(idx) if idx < self.n_pods else self.svc_uid(idx - self.n_pods)

    def k8s_ops(self) -> List[Tuple[str, str, str, str]]:
        """(kind, event_type, uid, ip) in ADD order: all pods, then all services — so node ids are
        pods 0..P-1, services P..P+S-1."""
        ops = [("pod", "ADD", self.pod_uid(i), ip_str(int(self.pod_ips[i]))) for i in range(self.n_pods)]
        ops += [("svc", "ADD", self.svc_uid(j), ip_str(int(self.svc_ips[j]))) for j in range(self.n_svcs)]
        return ops


def make_topology(pods: int, edges: int, seed: int, svcs: int | None = None) -> Topology:
    P = pods
    S = svcs if svcs is not None else max(1, P // 2)      # testconfig/config1.json:4-5 ratio
    N = P + S
    E = edges
    assert E <= P * (N - 1), "more edges than distinct (src, dst) pairs"
    pod_ips = (POD_IP_BASE + np.arange(P, dtype=np.uint64)).astype(np.uint32)
    svc_ips = (SVC_IP_BASE + np.arange(S, dtype=np.uint64)).astype(np.uint32)

    # out-degree ~ Pareto(alpha=1.2), clipped to [1, N/4], rescaled to sum E
    u = uniform01(seed, P, 1)
    dmax = max(1, N // 4)
    raw = np.minimum(u ** (-1.0 / 1.2), float(dmax))
    deg = np.maximum(1, np.floor(raw * (E / raw.sum()))).astype(np.int64)
    deg = np.minimum(deg, dmax)
    # fix the sum to exactly E, spreading the remainder over pods in a fixed pseudo-random order
    order = np.argsort(splitmix64(seed, P, 2), kind="stable")
    diff = int(E - deg.sum())
    guard = 0
    while diff != 0 and guard < 64:
        guard += 1
        step = 1 if diff > 0 else -1
        room = (deg[order] < dmax) if step > 0 else (deg[order] > 1)
        idx = order[room][: abs(diff)]
        if len(idx) == 0:
            # everything saturated: relax the clip
            dmax += 1
            continue
        deg[idx] += step
        diff = int(E - deg.sum())
    assert deg.sum() == E, (deg.sum(), E)

    svc_cdf = _zipf_cdf(S, 1.0)
    pod_cdf = _zipf_cdf(P, 1.0)
    # popularity rank -> node, through fixed permutations (hot services are not just index 0,1,2…)
    svc_perm = np.argsort(splitmix64(seed, S, 3), kind="stable")
    pod_perm = np.argsort(splitmix64(seed, P, 4), kind="stable")

    src_all = np.repeat(np.arange(P, dtype=np.int64), deg)
    have = np.zeros(0, dtype=np.uint64)         # sorted unique keys src<<32|dst
    need = deg.copy()
    rnd = 0
    while need.sum() > 0:
        rnd += 1
        over = need + (need >> 1) + 4 if rnd > 1 else need
        over = np.where(need > 0, over, 0)
        src = np.repeat(np.arange(P, dtype=np.int64), over)
        m = len(src)
        u_kind = uniform01(seed, m, 10 + 3 * rnd)
        u_pick = uniform01(seed, m, 11 + 3 * rnd)
        to_svc = u_kind < 0.8
        dst = np.where(to_svc, P + svc_perm[np.minimum(np.searchsorted(svc_cdf, u_pick), S - 1)],
                       pod_perm[np.minimum(np.searchsorted(pod_cdf, u_pick), P - 1)]).astype(np.int64)
        ok = dst != src
        if rnd > 24:   # hubs fighting the Zipf head: fall back to uniform destinations
            dst = np.where(ok, dst, (src + 1) % N)
            uni = (splitmix64(seed, m, 12 + 3 * rnd) % np.uint64(N)).astype(np.int64)
            dst = np.where(u_kind < 0.5, uni, dst)
            ok = dst != src
        keys = (src[ok].astype(np.uint64) << np.uint64(32)) | dst[ok].astype(np.uint64)
        # keep draw order within a src so that the first `need` distinct new ones win
        _, first = np.unique(keys, return_index=True)
        keys = keys[np.sort(first)]
        if len(have):
            keys = keys[~np.isin(keys, have, assume_unique=False)]
        ksrc = (keys >> np.uint64(32)).astype(np.int64)
        # rank of each key within its src (draw order)
        o = np.argsort(ksrc, kind="stable")
        ksrc_s = ksrc[o]
        start = np.searchsorted(ksrc_s, np.arange(P))
        rank = np.arange(len(o)) - start[ksrc_s]
        take = rank < need[ksrc_s]
        newk = keys[o][take]
        need -= np.bincount(ksrc_s[take], minlength=P)
        have = np.union1d(have, newk)
        assert rnd < 200, "edge sampling did not converge"
    assert len(have) == E
    edge_src = (have >> np.uint64(32)).astype(np.int64)
    edge_dst = (have & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.array_equal(np.sort(src_all), np.sort(edge_src))
    return Topology(P, S, pod_ips, svc_ips, edge_src, edge_dst, seed)


# ------------------------------------------------------------------------------------------------
# events
# ------------------------------------------------------------------------------------------------
EXTERNAL_HOSTS = [f"api{k}.example-{k % 7}.com" for k in range(64)]


def make_events(topo: Topology, n_events: int, seed: int, *, mixed: bool = False, stream_base: int = 100,
                t0_ns: int = 1_000_000_000, with_raw_outbound: bool = False,
                with_reverse: bool = False, fixed_labels: bool = False) -> Tuple[np.ndarray, List[str]]:
    """Packed events for ``topo`` (SURVEY.md §8d "Events").  Returns (events, labels) where
    labels[i] is the Host header of host_label i+1.

    mixed=False : HTTP only (configs 1-4).  mixed=True: 70 % HTTP / 15 % KAFKA / 15 % POSTGRES (config 5).
    fixed_labels: label ids are the indices of EXTERNAL_HOSTS (+1) instead of first-use order.
    with_raw_outbound / with_reverse add the edge cases the parity tests exercise (raw-IP outbound
    destinations; AMQP DELIVER / Redis PUSHED_EVENT direction reversal)."""
    E = len(topo.edge_src)
    P = topo.n_pods
    sb = stream_base
    cdf = _zipf_cdf(E, 0.8)
    perm = np.argsort(splitmix64(seed, E, sb + 0), kind="stable")
    e_idx = perm[np.minimum(np.searchsorted(cdf, uniform01(seed, n_events, sb + 1)), E - 1)]

    anomalous = uniform01(topo.seed, E, 5) < 0.01              # 1 % of edges, a topology property
    # lognormal(mu=ln 5e6, sigma=0.8) via Box-Muller
    u1 = uniform01(seed, n_events, sb + 2); u2 = uniform01(seed, n_events, sb + 3)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    dur = np.exp(np.log(5e6) + 0.8 * z)
    an = anomalous[e_idx]
    dur = np.where(an, dur * 20.0, dur)
    dur = np.maximum(1.0, np.rint(dur)).astype(np.uint64)

    us = uniform01(seed, n_events, sb + 4)
    status = np.where(us < 0.97, 200, np.where(us < 0.99, 404, 503)).astype(np.uint16)
    status = np.where(an & (uniform01(seed, n_events, sb + 5) < 0.30), 503, status).astype(np.uint16)

    ev = np.zeros(n_events, dtype=EVENT_DTYPE)
    ev["saddr"] = topo.pod_ips[topo.edge_src[e_idx]]
    ev["daddr"] = topo.node_ip(topo.edge_dst[e_idx])
    ev["status"] = status
    ev["protocol"] = PROTO_HTTP
    ev["duration_ns"] = dur
    jitter = (splitmix64(seed, n_events, sb + 6) % np.uint64(1000)).astype(np.uint64)
    ev["write_time_ns"] = np.uint64(t0_ns) + np.arange(n_events, dtype=np.uint64) * np.uint64(200) + jitter

    if mixed:
        up = uniform01(seed, n_events, sb + 7)
        kafka = (up >= 0.70) & (up < 0.85)
        pg = up >= 0.85
        ev["protocol"] = np.where(kafka, PROTO_KAFKA, np.where(pg, PROTO_POSTGRES, PROTO_HTTP))
        # KAFKA: status 1; POSTGRES: 1 ok / 2 error (ebpf/c/postgres.c:90-91); errors follow the 5xx draw
        st = ev["status"].copy()
        st = np.where(kafka, 1, st)
        st = np.where(pg, np.where(status >= 500, 2, 1), st)
        ev["status"] = st.astype(np.uint16)

    uk = uniform01(seed, n_events, sb + 8)
    unk_src = uk < 0.005                                        # unknown saddr: dropped (data.go:829-832)
    ev["saddr"] = np.where(unk_src, UNKNOWN_SRC_BASE + (splitmix64(seed, n_events, sb + 9) % np.uint64(251)).astype(np.uint32), ev["saddr"])
    outb = (uk >= 0.005) & (uk < 0.010)                         # unknown daddr + Host header: outbound
    k = (splitmix64(seed, n_events, sb + 10) % np.uint64(len(EXTERNAL_HOSTS))).astype(np.uint32)
    ev["daddr"] = np.where(outb, EXTERNAL_IP_BASE + k, ev["daddr"])
    http = ev["protocol"] == PROTO_HTTP
    labels: List[str] = []
    if outb.any():
        # label ids in first-use order, as the host packer interns them
        use = np.flatnonzero(outb & http)
        if fixed_labels:      # one global label table (sharded feeders must agree on the ids)
            order = np.arange(len(EXTERNAL_HOSTS))
        else:
            _, first = np.unique(k[use], return_index=True)
            order = k[use][np.sort(first)]
        lab_of = np.zeros(len(EXTERNAL_HOSTS), dtype=np.uint32)
        lab_of[order] = np.arange(1, len(order) + 1, dtype=np.uint32)
        labels = [EXTERNAL_HOSTS[int(x)] for x in order]
        ev["host_label"] = np.where(outb & http, lab_of[k], 0)
    if with_raw_outbound:
        ur = uniform01(seed, n_events, sb + 11)
        raw = (ur < 0.004) & ~unk_src
        kk = (splitmix64(seed, n_events, sb + 12) % np.uint64(97)).astype(np.uint32)
        ev["daddr"] = np.where(raw, EXTERNAL_IP_BASE + 0x100 + kk * np.uint32(7), ev["daddr"])
        ev["host_label"] = np.where(raw, 0, ev["host_label"])
    if with_reverse:
        uv = uniform01(seed, n_events, sb + 13)
        amqp = uv < 0.01
        redis = (uv >= 0.01) & (uv < 0.02)
        ev["protocol"] = np.where(amqp, PROTO_AMQP, np.where(redis, PROTO_REDIS, ev["protocol"]))
        ev["flags"] = np.where(amqp | redis, ev["flags"] | EV_REVERSE, ev["flags"])
        ev["status"] = np.where(amqp, 1, np.where(redis, np.where(status >= 500, 2, 1), ev["status"])).astype(np.uint16)
        ev["host_label"] = np.where(amqp | redis, 0, ev["host_label"])
    ut = uniform01(seed, n_events, sb + 14)
    ev["flags"] = np.where((ut < 0.2) & (ev["protocol"] == PROTO_HTTP), ev["flags"] | EV_TLS, ev["flags"])
    return ev, labels


def make_config(config: int, *, events: int | None = None, seed: int | None = None):
    """(topology, events, labels, layers) of one BASELINE.json config."""
    c = CONFIGS[config]
    seed = SEED_BASE + config if seed is None else seed
    topo = make_topology(c["pods"], c["edges"], seed)
    ev, labels = make_events(topo, events if events is not None else c["events"], seed, mixed=(config == 5))
    return topo, ev, labels, c["layers"]


# ------------------------------------------------------------------------------------------------
# full wire records (struct l7_event, 1096 B) — what the reference's perf reader consumes
# ------------------------------------------------------------------------------------------------
_HTTP_METHOD_GET = 1


def to_wire(events: np.ndarray, labels: List[str], *, pid: int = 4242) -> bytes:
    """bpfL7Event records (ebpf/l7_req/l7.go:345-369) equivalent to ``events``.

    HTTP payloads look like the simulator's ("GET /user HTTP1.1", main_benchmark_test.go:562) plus
    a Host header; POSTGRES events are simple queries; KAFKA/AMQP/REDIS carry minimal payloads.
    The wire status field is 32-bit; packed status is its 16-bit saturation."""
    n = len(events)
    buf = np.zeros((n, L7_WIRE_SIZE), dtype=np.uint8)
    v64 = lambda a: np.ascontiguousarray(a, dtype="<u8").view(np.uint8).reshape(n, 8)
    v32 = lambda a: np.ascontiguousarray(a, dtype="<u4").view(np.uint8).reshape(n, 4)
    buf[:, 0:8] = v64(np.arange(n) % 97 + 3)                    # fd
    buf[:, 8:16] = v64(events["write_time_ns"])
    buf[:, 16:20] = v32(np.full(n, pid))
    buf[:, 20:24] = v32(events["status"].astype(np.uint32))
    buf[:, 24:32] = v64(events["duration_ns"])
    buf[:, 32] = events["protocol"]
    proto = events["protocol"]
    rev = (events["flags"] & EV_REVERSE) != 0
    method = np.zeros(n, dtype=np.uint8)
    method[proto == PROTO_HTTP] = _HTTP_METHOD_GET
    method[proto == PROTO_POSTGRES] = 2                          # SIMPLE_QUERY
    method[proto == PROTO_KAFKA] = np.where((events["flags"][proto == PROTO_KAFKA] & EV_CONSUME) != 0, 2, 1)
    method[proto == PROTO_AMQP] = np.where(rev[proto == PROTO_AMQP], 2, 1)   # DELIVER / PUBLISH
    method[proto == PROTO_REDIS] = np.where(rev[proto == PROTO_REDIS], 2, 1)  # PUSHED_EVENT / COMMAND
    buf[:, 33] = method
    buf[:, 1066] = (events["flags"] & EV_TLS) != 0
    buf[:, 1076:1080] = v32(events["saddr"])
    buf[:, 1080:1082] = np.ascontiguousarray(32768 + (np.arange(n) % 28232), dtype="<u2").view(np.uint8).reshape(n, 2)
    buf[:, 1084:1088] = v32(events["daddr"])
    dport = np.where(proto == PROTO_POSTGRES, 5432, np.where(proto == PROTO_KAFKA, 9092, 80))
    buf[:, 1088:1090] = np.ascontiguousarray(dport, dtype="<u2").view(np.uint8).reshape(n, 2)
    for i in range(n):
        p = int(proto[i])
        if p == PROTO_HTTP:
            lab = int(events["host_label"][i])
            external = EXTERNAL_IP_BASE <= int(events["daddr"][i]) < EXTERNAL_IP_BASE + 0x10000
            if lab:
                hdr = f"Host: {labels[lab - 1]}\r\n"
            elif external:
                hdr = ""                      # raw-IP outbound: no Host header (data.go:855-864)
            else:
                hdr = "Host: svc.cluster.local\r\n"   # in-cluster call; never used as a UID
            pl = f"GET /user HTTP1.1\r\n{hdr}Accept: */*\r\n\r\n".encode()
        elif p == PROTO_POSTGRES:
            q = b"SELECT * FROM users WHERE id = 1\x00"
            pl = b"Q" + (len(q) + 4).to_bytes(4, "big") + q
        elif p == PROTO_REDIS:
            pl = b"*2\r\n$3\r\nGET\r\n$3\r\nkey\r\n"
        else:
            pl = b""
        buf[i, 36:36 + len(pl)] = np.frombuffer(pl, dtype=np.uint8)
        buf[i, 1060:1064] = np.frombuffer(len(pl).to_bytes(4, "little"), dtype=np.uint8)
        buf[i, 1064] = 1
    return buf.tobytes()
