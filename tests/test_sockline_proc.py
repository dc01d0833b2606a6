"""f-2, the rest of the socket-line source: a new line seeded from the proc file system and process exit.

NewSocketLine(fetch = true) -> getConnectionInfo (aggregator/sock_num_line.go:38-54, 351-429): the fd's link names a
socket inode, the FIRST line of /proc/<pid>/net/tcp that contains the inode's digits is parsed for the address pair, the
line's values are cleared and one open value stamped convertUserTimeToKernelTime(time.Now()) (data.go:1745-1747) added;
the event that caused the creation is applied afterwards (the re-queue of data.go:430-450).  processExit
(data.go:363-398) drops the process' socket map (cluster.go:97-110), its HTTP/2 parsers and Postgres statements by the
string prefix of the decimal pid — and no MySQL statement (the loop ranges over the wrong map).

The reference has no test of either; what it holds is the worked example in a source comment
(sock_num_line.go:244-246), used here as the known answer.  The oracle (oracle/sockline.c) and the product
(alaz_amd/csrc/host/sockline.cpp) are written separately and compared on proc trees built under tmp_path.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle
from oracle.pyoracle import SockLine
from tests.test_sockline import _ip, _tcp_wire

HEADER = "  sl  local_address rem_address   st tx_queue rx_queue tr tm->when retrnsmt   uid  timeout inode"
REF_LINE = "   0: 7038A8C0:A24A C28D640A:0050 01 00000000:00000000 02:000002E0 00000000     0        0 5276530 2 ffff8e8be7a0bd40 20 4 24 10 -1"


def _hexaddr(ip, port):
    a, b, c, d = (int(x) for x in ip.split("."))
    return "%02X%02X%02X%02X:%04X" % (d, c, b, a, port)


def _tcp_row(sl, lip, lport, rip, rport, inode, state="01"):
    return "%4d: %s %s %s 00000000:00000000 00:00000000 00000000  1000        0 %s 1 0000000000000000 20 4 30 10 -1" % (
        sl, _hexaddr(lip, lport), _hexaddr(rip, rport), state, inode)


def _mkproc(root, pid, fds=None, rows=None, header=True):
    """fds: {fd: link text}; rows: the lines of net/tcp below the header (None: no file)"""
    d = os.path.join(root, str(pid))
    os.makedirs(os.path.join(d, "fd"), exist_ok=True)
    os.makedirs(os.path.join(d, "net"), exist_ok=True)
    for fd, link in (fds or {}).items():
        p = os.path.join(d, "fd", str(fd))
        if os.path.lexists(p):
            os.unlink(p)
        os.symlink(link, p)                                   # dangling on purpose: readlink returns the text
    if rows is not None:
        with open(os.path.join(d, "net", "tcp"), "w") as f:
            f.write("\n".join(([HEADER] if header else []) + list(rows)) + "\n")


def _host():
    from alaz_amd import build, hostlib
    build.build_all()
    return hostlib


def _vals_cpp(line):
    return line.values()


def _vals_orc(line):
    return [(t, lm, None if si is None else (_ip(si[0]), si[1], _ip(si[2]), si[3])) for t, lm, si in line.values()]


# ---- known answers --------------------------------------------------------------------------------------------------
def test_kat_the_references_own_example_line():
    """sock_num_line.go:244-246: "7038A8C0:A24A C28D640A:0050" is 192.168.56.112:41546 -> 10.100.141.194:80"""
    assert pyoracle.parse_tcp_line(REF_LINE) == ("192.168.56.112", 41546, "10.100.141.194", 80)
    assert _host().proc_parse_tcp_line(REF_LINE) == (_ip("192.168.56.112"), 41546, _ip("10.100.141.194"), 80)


@pytest.mark.parametrize("link,inode", [
    ("socket:[5276530]", "5276530"), ("socket:[0]", "0"), ("pipe:[123]", None), ("anon_inode:[eventpoll]", None),
    ("/dev/null", None), ("socket:[]", None), ("socket:[12a]", None), ("socket:[x] socket:[77]", "77"),
    ("xsocket:[9]", "9"), ("socket:[1][2]", "1"), ("", None)])
def test_kat_inode_of_a_link(link, inode):
    """getInodeFromFD's regexp `socket:\\[(\\d+)\\]` (sock_num_line.go:358-363): first match anywhere in the text"""
    assert pyoracle.inode_from_link(link) == inode
    assert _host().proc_inode_of_link(link) == inode


@pytest.mark.parametrize("line", [
    REF_LINE, HEADER, "0: 0100007F:1F90 00000000:0000 0A", "0: ZZ38A8C0:A24A C28D640A:0050 01", "0: 7038A8C0:FFFFF C28D640A:-050 01",
    "0: 7038A8C0:A2_A C28D640A: 01", "x y", "", "0: 7038A8C0 C28D640A:0050", "0:\t7038a8c0:a24a\tc28d640a:0050\t01",
    "0: 7038A8C0:FFFFFFFFFFFFFFFFFF C28D640A:+50 01", "0: 7038A8C0:- C28D640A:+ 01", "0: 7038Aa-8C0:A24A C2+D640A:0050 01"])
def test_tcp_line_parsing_equals_the_oracle_on_odd_lines(line):
    """ignored ParseInt errors read as 0, ports beyond 65535 as 0, short columns are refused (the reference would panic)"""
    a = pyoracle.parse_tcp_line(line)
    b = _host().proc_parse_tcp_line(line)
    assert (a is None) == (b is None)
    if a is not None:
        # (the reference prints "%d" of a SIGNED parse: the pair "-8" becomes the text "-8"; the product's numeric address keeps its low byte)
        low = lambda s: sum((int(x) & 255) << (24 - 8 * i) for i, x in enumerate(s.split(".")))
        assert (low(a[0]), a[1], low(a[2]), a[3]) == b


# ---- getConnectionInfo ------------------------------------------------------------------------------------------------
def test_seed_replaces_the_lines_values_by_one_open_value(tmp_path):
    hl = _host(); root = str(tmp_path)
    _mkproc(root, 4242, {7: "socket:[5276530]"}, [_tcp_row(0, "10.0.0.9", 1, "10.0.0.8", 2, 11), REF_LINE])
    o, c = SockLine(4242, 7), hl.SocketLine(4242, 7)
    o.add(5, pyoracle.sockinfo("1.1.1.1", 1, "2.2.2.2", 2)); c.add(5, (_ip("1.1.1.1"), 1, _ip("2.2.2.2"), 2))
    assert o.seed_from_proc(root, 777) == c.seed_from_proc(root, 777) == 0
    assert _vals_orc(o) == _vals_cpp(c) == [(777, 0, (_ip("192.168.56.112"), 41546, _ip("10.100.141.194"), 80))]


def test_seed_failures_leave_the_line_alone(tmp_path):
    hl = _host(); root = str(tmp_path)
    _mkproc(root, 1, {3: "pipe:[99]", 4: "socket:[424242]", 5: "socket:[5276530]"}, [REF_LINE])
    _mkproc(root, 2, {3: "socket:[5276530]"}, None)                              # no net/tcp
    _mkproc(root, 3, {3: "socket:[5276530]"}, ["0: 7038A8C0 5276530"])           # a line that cannot be indexed
    cases = [(1, 9, 1), (1, 3, 2), (2, 3, 3), (1, 4, 4), (3, 3, 5), (77, 3, 1), (1, 5, 0)]
    for pid, fd, want in cases:
        o, c = SockLine(pid, fd), hl.SocketLine(pid, fd)
        o.add(5, pyoracle.sockinfo("1.1.1.1", 1, "2.2.2.2", 2)); c.add(5, (_ip("1.1.1.1"), 1, _ip("2.2.2.2"), 2))
        assert o.seed_from_proc(root, 9) == c.seed_from_proc(root, 9) == want, (pid, fd)
        assert _vals_orc(o) == _vals_cpp(c)
        assert (len(o) == 1 and o.values()[0][0] == (9 if want == 0 else 5))


def test_the_inode_is_matched_as_a_substring_of_the_whole_line(tmp_path):
    """findTCPConnection (sock_num_line.go:368-382) is strings.Contains on the line: inode 80 is found in the PORT column
    of an earlier connection, inode 1000 in the uid column, and a short inode in a longer one"""
    hl = _host(); root = str(tmp_path)
    rows = [_tcp_row(0, "10.0.0.1", 0x8080, "10.0.0.2", 0x1234, 555),            # hex text "8080" contains "80"
            _tcp_row(1, "10.0.0.3", 0x1111, "10.0.0.4", 0x2222, 80),
            _tcp_row(2, "10.0.0.5", 0x3333, "10.0.0.6", 0x4444, 1000),             # the uid column of every row says 1000
            _tcp_row(3, "10.0.0.7", 0x5555, "10.0.0.8", 0x6666, 5559)]
    _mkproc(root, 9, {3: "socket:[80]", 4: "socket:[1000]", 5: "socket:[555]", 6: "socket:[5559]"}, rows)
    want = {3: ("10.0.0.1", 0x8080, "10.0.0.2", 0x1234), 4: ("10.0.0.1", 0x8080, "10.0.0.2", 0x1234),
            5: ("10.0.0.1", 0x8080, "10.0.0.2", 0x1234), 6: ("10.0.0.7", 0x5555, "10.0.0.8", 0x6666)}
    for fd, (lip, lp, rip, rp) in want.items():
        o, c = SockLine(9, fd), hl.SocketLine(9, fd)
        assert o.seed_from_proc(root, 1) == c.seed_from_proc(root, 1) == 0
        assert _vals_orc(o) == _vals_cpp(c) == [(1, 0, (_ip(lip), lp, _ip(rip), rp))], fd


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_proc_trees(tmp_path, seed):
    """random link texts and net/tcp files (missing header, CRLF, blank and truncated rows, listening sockets)"""
    hl = _host(); root = str(tmp_path); rng = np.random.default_rng(seed)
    for pid in range(1, 9):
        inodes = [int(x) for x in rng.integers(1, 3000, size=6)]
        rows = []
        for k, ino in enumerate(inodes):
            r = _tcp_row(k, "10.%d.%d.%d" % tuple(rng.integers(0, 256, 3)), int(rng.integers(0, 65536)),
                         "172.%d.%d.%d" % tuple(rng.integers(0, 256, 3)), int(rng.integers(0, 65536)), ino, state=("0A" if rng.random() < 0.2 else "01"))
            u = rng.random()
            rows.append(r[:int(rng.integers(0, 30))] if u < 0.1 else ("" if u < 0.15 else (r + "\r" if u < 0.3 else r)))
        links = {}
        for fd in range(3, 12):
            u = rng.random()
            ino = inodes[int(rng.integers(0, 6))] if u < 0.7 else int(rng.integers(1, 99))
            links[fd] = ("socket:[%d]" % ino) if rng.random() < 0.85 else ["pipe:[%d]" % ino, "/dev/pts/0", "socket:[%d" % ino][int(rng.integers(0, 3))]
        _mkproc(root, pid, links, rows if rng.random() < 0.9 else None, header=rng.random() < 0.8)
    codes = set()
    for pid in range(1, 10):
        for fd in range(3, 13):
            o, c = SockLine(pid, fd), hl.SocketLine(pid, fd)
            a, b = o.seed_from_proc(root, 123), c.seed_from_proc(root, 123)
            assert a == b, (pid, fd)
            assert _vals_orc(o) == _vals_cpp(c), (pid, fd)
            codes.add(a)
    assert {0, 1, 4} <= codes


# ---- processTcpConnect with seeding; processExit ----------------------------------------------------------------------
def _pair(o, g, pid, fd):
    a, b = g.sockline(pid, fd), o.sockline(pid, fd)
    assert (a is None) == (b is None), (pid, fd)
    return (None, None) if a is None else (_vals_cpp(a), _vals_orc(b))


def test_tracker_seeds_a_new_line_before_the_requeued_event_reaches_it(tmp_path):
    from alaz_amd import engine
    hl = _host(); root = str(tmp_path)
    E, Cl = pyoracle.TCP_ESTABLISHED, pyoracle.TCP_CLOSED
    FK, FU, NOW = 5_000_000, 1_700_000_000_000_000_000, 1_700_000_000_000_020_000         # -> the seeded stamp is 5 020 000
    _mkproc(root, 10, {3: "socket:[111]", 4: "socket:[222]", 5: "socket:[333]", 6: "pipe:[1]"},
            [_tcp_row(0, "10.0.0.1", 40000, "10.96.0.5", 80, 111),               # same pair as the event of fd 3
             _tcp_row(1, "10.0.0.1", 40001, "10.0.0.2", 8080, 222),              # fd 4: closed again before the seed's stamp
             _tcp_row(2, "10.0.0.1", 40002, "10.96.0.7", 443, 333)])             # fd 5: /proc already shows ANOTHER peer
    o = pyoracle.Oracle(FK, FU, log_limit=100)
    cfg = engine.make_config(max_known_nodes=64, max_edges=1024, max_labels=64)
    g = hl.GraphDS(cfg, engine_lib=None, batch=1)
    o.set_proc_root(root, NOW); g.set_proc_root(root, FK, FU, NOW)
    for i in (1, 2):
        o.pod("ADD", f"pod-{i}", f"10.0.0.{i}"); g.PersistPod(f"pod-{i}", f"10.0.0.{i}")
    for i in (5, 7):
        o.svc("ADD", f"svc-{i}", f"10.96.0.{i}"); g.PersistService(f"svc-{i}", f"10.96.0.{i}")
    wire = _tcp_wire([
        (E, 10, 3, 5_000_100, "10.0.0.1", 40000, "10.96.0.5", 80),
        (E, 10, 4, 5_000_200, "10.0.0.1", 40001, "10.0.0.2", 8080),
        (Cl, 10, 4, 5_000_300, "10.0.0.1", 40001, "10.0.0.2", 8080),
        (E, 10, 5, 5_000_400, "10.0.0.1", 40002, "10.96.0.5", 80),
        (E, 10, 6, 5_000_500, "10.0.0.1", 40003, "10.0.0.2", 80),              # link is a pipe: the line starts empty
        (E, 11, 3, 5_000_600, "10.0.0.2", 40004, "10.0.0.1", 80),              # no such process under the root
        (E, 10, 3, 5_000_700, "10.0.0.1", 40000, "10.96.0.5", 80),             # the line exists by now: no second seed
    ])
    assert g.tcp_wire(wire) == o.tcp_wire(wire) == 7
    assert g.seed_stats() == (3, 2)
    S = 5_020_000
    a, b = _pair(o, g, 10, 3)
    # the event's pair equals the seed's: AddValue's last-equal rule drops it — the line holds the seed only
    assert a == b == [(S, 0, (_ip("10.0.0.1"), 40000, _ip("10.96.0.5"), 80))]
    a, b = _pair(o, g, 10, 4)
    # the seed is stamped AFTER the close: the line ends on an open value (a reference quirk worth knowing: it reports alive)
    assert a == b == [(5_000_300, 0, None), (S, 0, (_ip("10.0.0.1"), 40001, _ip("10.0.0.2"), 8080))]
    a, b = _pair(o, g, 10, 5)
    assert a == b == [(5_000_400, 0, (_ip("10.0.0.1"), 40002, _ip("10.96.0.5"), 80)), (S, 0, (_ip("10.0.0.1"), 40002, _ip("10.96.0.7"), 443))]
    a, b = _pair(o, g, 10, 6)
    assert a == b == [(5_000_500, 0, (_ip("10.0.0.1"), 40003, _ip("10.0.0.2"), 80))]
    a, b = _pair(o, g, 11, 3)
    assert a == b and len(a) == 1
    assert g.sweep(99) == 5 and o.sweep(99) == 5
    rows = sorted((r[4], r[5], r[7]) for r in o.alive_rows())
    assert rows == [(40000, "10.96.0.5", "svc-5"), (40001, "10.0.0.2", "pod-2"), (40002, "10.96.0.7", "svc-7"), (40003, "10.0.0.2", "pod-2"), (40004, "10.0.0.1", "pod-1")]
    from alaz_amd import replay
    ev = g.mock_events(); al = ev[(ev["flags"] & replay.EV_ALIVE) != 0]
    assert sorted((int(e["saddr"]), int(e["daddr"])) for e in al) == sorted((_ip(r[1]), _ip(r[5])) for r in o.alive_rows())


def test_process_exit_forgets_the_process_lines_and_statements_by_pid_prefix():
    """data.go:363-398 + cluster.go:97-110.  pid 12 exits: its lines go, those of 123 stay (SocketMaps is indexed by the
    number) — but the Postgres statements and HTTP/2 parsers of 123 go too (string prefix "12")."""
    from alaz_amd import engine
    from tests.test_oracle_golden import _wire
    hl = _host()
    E = pyoracle.TCP_ESTABLISHED
    o = pyoracle.Oracle(0, 0, log_limit=100)
    cfg = engine.make_config(max_known_nodes=64, max_edges=1024, max_labels=64)
    g = hl.GraphDS(cfg, engine_lib=None, batch=1)
    for i in (1, 2):
        o.pod("ADD", f"pod-{i}", f"10.0.0.{i}"); g.PersistPod(f"pod-{i}", f"10.0.0.{i}")
    wire = _tcp_wire([(E, 12, 3, 100, "10.0.0.1", 40000, "10.0.0.2", 80), (E, 12, 4, 110, "10.0.0.1", 40001, "10.0.0.2", 80),
                      (E, 123, 3, 120, "10.0.0.1", 40002, "10.0.0.2", 80), (E, 7, 3, 130, "10.0.0.2", 40003, "10.0.0.1", 80)])
    assert g.tcp_wire(wire) == o.tcp_wire(wire) == 4

    def parse(pid, fd, name):
        body = name + b"\x00" + b"SELECT 1\x00\x00\x00"
        r = bytearray(_wire(0x0A000001, 0x0A000002, proto=3, method=3, status=1, payload=b"P" + (len(body) + 4).to_bytes(4, "big") + body))
        r[0:8] = fd.to_bytes(8, "little"); r[16:20] = pid.to_bytes(4, "little")
        return bytes(r)
    pg = parse(12, 3, b"a") + parse(123, 3, b"a") + parse(7, 3, b"a") + parse(71, 3, b"a")
    assert o.l7_wire(pg) == 4
    g.ingest_wire(pg)
    assert o.pg_stmt_count() == g.pg_statements() == 4
    o.process_exit(12); g.proc_exit(12)
    assert o.sockline_count() == g.sockline_count() == 2
    assert o.sockline(12, 3) is None and g.sockline(12, 3) is None and o.sockline(12, 4) is None and g.sockline(12, 4) is None
    assert len(o.sockline(123, 3)) == len(g.sockline(123, 3)) == 1
    assert o.pg_stmt_count() == g.pg_statements() == 2                           # "7-3-a" and "71-3-a"
    assert g.sweep(5) == o.sweep(5) == 2
    o.process_exit(12); g.proc_exit(12)                                          # a second exit of the same pid: nothing left to do
    o.process_exit(7); g.proc_exit(7)
    assert o.sockline_count() == g.sockline_count() == 1 and o.pg_stmt_count() == g.pg_statements() == 0
    # the pid comes back (pid reuse): a fresh line
    w2 = _tcp_wire([(E, 12, 3, 200, "10.0.0.1", 40009, "10.0.0.2", 80)])
    assert g.tcp_wire(w2) == o.tcp_wire(w2) == 1
    assert _vals_cpp(g.sockline(12, 3)) == _vals_orc(o.sockline(12, 3)) == [(200, 0, (_ip("10.0.0.1"), 40009, _ip("10.0.0.2"), 80))]
