"""f-2: the socket-line state machine and the alive-connection source.

The oracle's restatement (oracle/sockline.c) is pinned by the reference's own known-answer tests,
re-encoded here: aggregator/sock_line_test.go TestSocketLine (:10-347), TestXxx2 (:443-476),
TestAlreadyEstablishCanBeFound (:478-501) and the retention loop of TestXxx (:349-441).  The C++ host
implementation (alaz_amd/csrc/host/sockline.*) is then checked against the oracle on random traces.
"""
import numpy as np
import pytest

from oracle import pyoracle
from oracle.pyoracle import SockLine, sockinfo


def test_kat_second_open_is_found_after_its_timestamp():
    """sock_line_test.go:443-476 (TestXxx2): AddValue(0, yy); AddValue(247453008321477, xx); GetValue(247453008321499) -> xx"""
    nl = SockLine(1, 0)
    nl.add(0, sockinfo(saddr="yy"))
    nl.add(247453008321477, sockinfo(saddr="xx"))
    s, err = nl.get(247453008321499)
    assert err is None and s.saddr == b"xx"


def test_kat_already_established_can_be_found():
    """sock_line_test.go:478-501: a single open at timestamp 0 answers GetValue(0)"""
    nl = SockLine(1, 0)
    nl.add(0, sockinfo(saddr="yy"))
    s, err = nl.get(0)
    assert err is None and s.saddr == b"yy"


def test_kat_identical_consecutive_opens_collapse_and_answer_later_queries():
    """sock_line_test.go:10-347 (TestSocketLine): hundreds of AddValue calls with an identical (empty) SockInfo
    keep only the first entry (last-equal de-duplication, sock_num_line.go:71-78); GetValue after the last
    timestamp returns it without error.  (First / last stamps and the queried stamp are the reference's; the
    ones in between are regenerated — they never reach the line.)"""
    nl = SockLine(0, 0)
    first, last, query = 33805065332163, 33815077484050, 33835107729129
    rng = np.random.default_rng(7)
    stamps = [first] + sorted(int(x) for x in rng.integers(first + 1, last, size=300)) + [last]
    for ts in stamps:
        nl.add(ts, sockinfo())
    assert len(nl) == 1 and nl.values()[0][0] == first
    s, err = nl.get(query)
    assert err is None and s is not None


def test_kat_retention_loop_of_the_reference_test():
    """sock_line_test.go:349-441 (TestXxx): opens at 10/30/50, closes at 20/40/60; GetValue(52) and
    GetValue(33) match the opens at 50 and 30.  DeleteUnused (sock_num_line.go:159-208) first drops the
    trailing close (its pairing loop never appends the last element), then removes (open, close) pairs whose
    open was last matched more than 5 minutes before the newest match."""
    nl = SockLine(1, 0)
    for ts, si in ((10, sockinfo()), (20, None), (30, sockinfo(saddr="b")), (40, None), (50, sockinfo(saddr="c")), (60, None)):
        nl.add(ts, si)
    assert [v[0] for v in nl.values()] == [10, 20, 30, 40, 50, 60]
    t_old, t_new = 1_000, 1_000 + 6 * 60 * 1_000_000_000
    s, err = nl.get(33, now_ns=t_old); assert err is None and s.saddr == b"b"
    s, err = nl.get(52, now_ns=t_new); assert err is None and s.saddr == b"c"
    nl.delete_unused()
    # pass 1 keeps [10, 20, 30, 40, 50] (60 dropped); pass 2: (30, 40) was matched 6 min before the newest match
    # -> removed; (10, 20) was never matched (LastMatch 0) -> removed as well
    assert [v[0] for v in nl.values()] == [50]


def test_getvalue_special_cases():
    """sock_num_line.go:83-157 branch by branch."""
    nl = SockLine(1, 3)
    assert nl.get(5) == (None, "sock line is empty")
    a = sockinfo("10.0.0.1", 1000, "10.0.0.2", 80); b = sockinfo("10.0.0.1", 1001, "10.0.0.2", 80); c = sockinfo("10.0.0.1", 1002, "10.0.0.9", 80)
    nl.add(100, None)
    assert nl.get(50) == (None, "no smaller value found")                  # index 0 is a close
    nl = SockLine(1, 3)
    nl.add(100, a); nl.add(200, None)
    s, err = nl.get(150); assert err is None and s.sport == 1000            # closest previous open
    s, err = nl.get(100); assert err is None and s.sport == 1000            # equal stamp: index 0 -> first value
    s, err = nl.get(200 + 59_000_000_000); assert err is None and s.sport == 1000   # after a closing last entry, < 1 minute after the open
    assert nl.get(100 + 60_000_000_000 + 1) == (None, "closed socket on last entry")
    nl.add(1000, b)                                                         # [100 open a, 200 close, 1000 open b], same daddr:dport
    s, err = nl.get(240); assert err is None and s.sport == 1000            # on the close: 240-100 < 1000-240 -> the previous open
    s, err = nl.get(600); assert err is None and s.sport == 1001            # 600-100 >= 1000-600 -> the next open
    nl = SockLine(1, 3)
    nl.add(100, a); nl.add(200, None); nl.add(300, c)
    assert nl.get(250) == (None, "closed socket")                           # neighbours go to different destinations
    # equal timestamps are inserted BEFORE the existing entry (lower bound, sock_num_line.go:311-322)
    nl = SockLine(1, 3)
    nl.add(100, a); nl.add(100, None)
    assert [v[2] is None for v in nl.values()] == [True, False]


def _tcp_wire(recs):
    """BpfTcpEvent records (ebpf/tcp_state/tcp.go:63-72)."""
    buf = np.zeros((len(recs), pyoracle.TCP_WIRE_SIZE), dtype=np.uint8)
    for i, (typ, pid, fd, ts, saddr, sport, daddr, dport) in enumerate(recs):
        r = buf[i]
        r[0:8] = np.frombuffer(np.uint64(fd).tobytes(), np.uint8); r[8:16] = np.frombuffer(np.uint64(ts).tobytes(), np.uint8)
        r[16:20] = np.frombuffer(np.uint32(typ).tobytes(), np.uint8); r[20:24] = np.frombuffer(np.uint32(pid).tobytes(), np.uint8)
        r[24:26] = np.frombuffer(np.uint16(sport).tobytes(), np.uint8); r[26:28] = np.frombuffer(np.uint16(dport).tobytes(), np.uint8)
        r[28:32] = [int(x) for x in saddr.split(".")]; r[44:48] = [int(x) for x in daddr.split(".")]
    return buf.tobytes()


def test_tcp_events_to_alive_connections():
    """processTcpConnect (data.go:404-506) + one clearSocketLines tick (:1628-1716): localhost filtered, a close
    without a line dropped, only lines whose last value is an open socket report, the source must be a pod,
    destination service-first / pod / ("outbound", ip)."""
    o = pyoracle.Oracle(0, 0, log_limit=100)
    o.pod("ADD", "pod-a", "10.0.0.1"); o.pod("ADD", "pod-b", "10.0.0.2"); o.svc("ADD", "svc-x", "10.96.0.5")
    E, Cl = pyoracle.TCP_ESTABLISHED, pyoracle.TCP_CLOSED
    wire = _tcp_wire([
        (E, 10, 3, 1000, "10.0.0.1", 40000, "10.96.0.5", 80),      # pod-a -> svc-x, stays open
        (E, 10, 4, 1100, "10.0.0.1", 40001, "10.0.0.2", 8080),     # pod-a -> pod-b, closed below
        (Cl, 10, 4, 1200, "10.0.0.1", 40001, "10.0.0.2", 8080),
        (E, 10, 5, 1300, "10.0.0.1", 40002, "93.184.216.34", 443), # pod-a -> outbound
        (E, 11, 3, 1400, "172.16.0.9", 40003, "10.0.0.2", 80),     # source is not a pod: ignored at sweep
        (E, 12, 3, 1500, "127.0.0.1", 40004, "10.0.0.2", 80),      # localhost: filtered
        (Cl, 13, 9, 1600, "10.0.0.2", 40005, "10.0.0.1", 80),      # close without a line: dropped
        (3, 14, 3, 1700, "10.0.0.2", 0, "0.0.0.0", 0),             # LISTEN: not handled by processTcpConnect
    ])
    assert o.tcp_wire(wire) == 5 and o.sockline_count() == 4
    assert o.sweep(now_ms=1_700_000_000_000) == 2
    rows = sorted(o.alive_rows(), key=lambda r: r[4])
    assert rows == [
        (1_700_000_000_000, "10.0.0.1", "pod", "pod-a", 40000, "10.96.0.5", "service", "svc-x", 80),
        (1_700_000_000_000, "10.0.0.1", "pod", "pod-a", 40002, "93.184.216.34", "outbound", "93.184.216.34", 443),
    ]
    # part 2: the alive connections are count-only edges of the open window
    from alaz_amd import weights
    o.window_close(weights.make_weights(1), 1)
    d = o.edge_dict()
    assert d[("pod", "pod-a", "service", "svc-x")][:5] == (0, 0, 0, 0, 0) and d[("pod", "pod-a", "service", "svc-x")][8] == 1
    assert d[("pod", "pod-a", "outbound", "93.184.216.34")][8] == 1 and len(d) == 2
    # DeleteUnused ran on every line: the (open, close) line lost its trailing close
    assert len(o.sockline(10, 4)) == 1


# ---- the C++ host implementation against the oracle ---------------------------------------------
def _ip(s):
    a, b, c, d = (int(x) for x in s.split("."))
    return (a << 24) | (b << 16) | (c << 8) | d


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_cpp_socket_line_equals_the_oracle_on_random_traces(seed):
    """AddValue / GetValue / DeleteUnused in random order, a handful of address pairs (so de-duplication,
    equal time stamps, closes between opens to the same destination and the 1-minute / 5-minute rules all
    occur): the C++ SocketLine (alaz_amd/csrc/host/sockline.cpp) must agree with the oracle after every step."""
    from alaz_amd import build, hostlib
    build.build_all()
    rng = np.random.default_rng(seed)
    cpp, orc = hostlib.SocketLine(7, 9), SockLine(7, 9)
    pairs = [("10.0.0.1", 1000 + k % 3, "10.0.0.%d" % (2 + k % 2), 80) for k in range(4)]
    now = 10**12
    for step in range(400):
        r = rng.random()
        ts = int(rng.integers(0, 50)) * 10**9 * (40 if rng.random() < 0.1 else 1)       # coarse stamps: ties happen
        if r < 0.45:
            p = pairs[int(rng.integers(0, len(pairs)))] if rng.random() < 0.7 else None
            cpp.add(ts, None if p is None else (_ip(p[0]), p[1], _ip(p[2]), p[3]))
            orc.add(ts, None if p is None else sockinfo(*p))
        elif r < 0.9:
            now += int(rng.integers(1, 200)) * 10**9
            a, ea = cpp.get(ts, now)
            b, eb = orc.get(ts, now)
            assert ea == eb, (step, ea, eb)
            if ea is None:
                assert a == (_ip(b.saddr.decode()), b.sport, _ip(b.daddr.decode()), b.dport)
        else:
            cpp.delete_unused(); orc.delete_unused()
        va = cpp.values()
        vb = [(t, lm, None if si is None else (_ip(si[0]), si[1], _ip(si[2]), si[3])) for t, lm, si in orc.values()]
        assert va == vb, step


def test_cpp_tracker_feeds_alive_records_to_the_engine_boundary():
    """BpfTcpEvent records -> ConnTracker::ProcessTcpConnect -> Sweep -> GraphDS::PersistAliveConnection ->
    SG_EV_ALIVE events at the C ABI (recording engine): same lines, same reported connections as the oracle."""
    from alaz_amd import build, engine, hostlib, replay
    build.build_all()
    E, Cl = pyoracle.TCP_ESTABLISHED, pyoracle.TCP_CLOSED
    rng = np.random.default_rng(11)
    recs = []
    for k in range(300):
        pid, fd = int(rng.integers(1, 6)), int(rng.integers(3, 12))
        s = "10.0.0.%d" % rng.integers(1, 5) if rng.random() < 0.9 else ("127.0.0.1" if rng.random() < 0.5 else "172.16.0.3")
        d = ("10.96.0.%d" % rng.integers(1, 4)) if rng.random() < 0.6 else ("10.0.0.%d" % rng.integers(1, 5) if rng.random() < 0.6 else "93.184.216.%d" % rng.integers(1, 5))
        recs.append((E if rng.random() < 0.6 else (Cl if rng.random() < 0.9 else 3), pid, fd, 1000 + 10 * k, s, 30000 + k, d, 80))
    wire = _tcp_wire(recs)
    o = pyoracle.Oracle(0, 0, log_limit=10_000)
    cfg = engine.make_config(max_known_nodes=64, max_edges=1024, max_labels=64)
    g = hostlib.GraphDS(cfg, engine_lib=None, batch=1)
    for i in range(1, 5):
        o.pod("ADD", f"pod-{i}", f"10.0.0.{i}"); g.PersistPod(f"pod-{i}", f"10.0.0.{i}")
    for i in range(1, 4):
        o.svc("ADD", f"svc-{i}", f"10.96.0.{i}"); g.PersistService(f"svc-{i}", f"10.96.0.{i}")
    assert g.tcp_wire(wire) == o.tcp_wire(wire)
    assert g.sockline_count() == o.sockline_count()
    for pid in range(1, 6):
        for fd in range(3, 12):
            a, b = g.sockline(pid, fd), o.sockline(pid, fd)
            assert (a is None) == (b is None)
            if a is not None:
                assert [(t, s is None) for t, _, s in a.values()] == [(t, s is None) for t, _, s in b.values()]
    n_lines_open = g.sweep(1234)
    o.sweep(1234)
    ev = g.mock_events()
    al = ev[(ev["flags"] & replay.EV_ALIVE) != 0]
    # the host reports every line whose last value is open; the join (source must be a pod ...) is the engine's
    assert len(al) == n_lines_open >= o.alive_count() > 0
    pod_ips = {_ip(f"10.0.0.{i}") for i in range(1, 5)}
    got = sorted((int(e["saddr"]), int(e["daddr"])) for e in al if int(e["saddr"]) in pod_ips)
    want = sorted((_ip(r[1]), _ip(r[5])) for r in o.alive_rows())
    assert got == want
    # DeleteUnused ran on both sides
    for pid in range(1, 6):
        for fd in range(3, 12):
            a, b = g.sockline(pid, fd), o.sockline(pid, fd)
            if a is not None:
                assert [(t, s is None) for t, _, s in a.values()] == [(t, s is None) for t, _, s in b.values()]


def test_tcp_close_forgets_the_connections_prepared_statements_by_key_prefix():
    """aggregator/data.go:496-503: on CLOSED every pgStmts key with the *string prefix* "pid-fd" is deleted — the keys are
    "pid-fd-name" (:1619-1621), so closing fd 7 also clears the statements of fds 70..79 of that pid.  Oracle and packer."""
    from alaz_amd import hostlib
    from tests.test_oracle_golden import _wire
    def parse(fd, name):                     # extended-query Parse message: 'P' len name\0 query\0 ...
        body = name + b"\x00" + b"SELECT * FROM t WHERE id = $1\x00\x00\x00"
        r = bytearray(_wire(0x0A000001, 0x0A000002, proto=3, method=3, status=1, payload=b"P" + (len(body) + 4).to_bytes(4, "big") + body))
        r[0:8] = fd.to_bytes(8, "little")
        return bytes(r)
    wire = parse(7, b"s1") + parse(7, b"s2") + parse(70, b"s1") + parse(8, b"s1")
    o = pyoracle.Oracle(0, 0); o.pod("ADD", "p1", "10.0.0.1"); o.pod("ADD", "p2", "10.0.0.2")
    pk = hostlib.Packer()
    assert o.l7_wire(wire) == 4 and len(pk.pack_wire(wire)) == 4
    assert o.pg_stmt_count() == pk.pg_statements() == 4
    o.tcp(5, 99, 7, 50, "10.0.0.1", 40000, "10.0.0.2", 5432)                      # CLOSED without a line: dropped, nothing forgotten
    assert o.pg_stmt_count() == 4
    o.tcp(1, 99, 7, 10, "10.0.0.1", 40000, "10.0.0.2", 5432); o.tcp(5, 99, 7, 60, "10.0.0.1", 40000, "10.0.0.2", 5432)
    pk.conn_closed(99, 7)
    assert o.pg_stmt_count() == pk.pg_statements() == 1                              # only "99-8-s1" is left
