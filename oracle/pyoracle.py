"""ctypes binding of oracle/libsgoracle.so.  TEST INFRASTRUCTURE ONLY (see sg_oracle.h):
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by alaz_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsgoracle.so")
OR_UID_MAX = 160
L7_WIRE_SIZE = 1096


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("sg_oracle.c", "sockline.c", "http2.c", "kafka.c", "lean_baseline.c", "sg_oracle.h")] + [os.path.join(_HERE, "..", "include", "servicegraph.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return LIB_PATH


class EdgeOut(C.Structure):
    _fields_ = [("sum_ns", C.c_uint64), ("max_ns", C.c_uint64), ("sumsq_us", C.c_uint64),
                ("from_ref", C.c_uint32), ("to_ref", C.c_uint32), ("count", C.c_uint32), ("err_count", C.c_uint32),
                ("score", C.c_float), ("lat_z", C.c_float), ("err_ratio", C.c_float), ("alive", C.c_uint32),
                ("p50_us", C.c_uint32), ("p99_us", C.c_uint32)]


class OrEdge(C.Structure):
    _fields_ = [("row", EdgeOut), ("from_type", C.c_char * 10), ("to_type", C.c_char * 10),
                ("from_uid", C.c_char * OR_UID_MAX), ("to_uid", C.c_char * OR_UID_MAX)]


class ReqInfo(C.Structure):
    _fields_ = [("start_time", C.c_int64), ("latency", C.c_uint64),
                ("from_ip", C.c_char * 16), ("from_type", C.c_char * 10), ("from_uid", C.c_char * OR_UID_MAX), ("from_port", C.c_uint16),
                ("to_ip", C.c_char * 16), ("to_type", C.c_char * 10), ("to_uid", C.c_char * OR_UID_MAX), ("to_port", C.c_uint16),
                ("protocol", C.c_char * 10), ("status_code", C.c_uint32), ("fail_reason", C.c_char * 4),
                ("method", C.c_char * 24), ("path", C.c_char * 1100), ("tls", C.c_uint8), ("is_kafka", C.c_uint8)]

    def as_tuple(self):
        """The 16 slots of ReqInfo in reference order (datastore/backend.go:824-839)."""
        d = lambda b: b.decode("latin-1")
        return (self.start_time, self.latency, d(self.from_ip), d(self.from_type), d(self.from_uid), self.from_port,
                d(self.to_ip), d(self.to_type), d(self.to_uid), self.to_port, d(self.protocol), self.status_code,
                d(self.fail_reason), d(self.method), d(self.path), bool(self.tls))


class SockInfo(C.Structure):
    """SockInfo, aggregator/socket.go:20-27"""
    _fields_ = [("pid", C.c_uint32), ("fd", C.c_uint64), ("saddr", C.c_char * 16), ("sport", C.c_uint16),
                ("daddr", C.c_char * 16), ("dport", C.c_uint16)]


class Alive(C.Structure):
    """datastore.AliveConnection, datastore/dto.go:96-106"""
    _fields_ = [("check_time", C.c_int64),
                ("from_ip", C.c_char * 16), ("from_type", C.c_char * 10), ("from_uid", C.c_char * OR_UID_MAX), ("from_port", C.c_uint16),
                ("to_ip", C.c_char * 16), ("to_type", C.c_char * 10), ("to_uid", C.c_char * OR_UID_MAX), ("to_port", C.c_uint16)]

    def as_tuple(self):
        d = lambda b: b.decode("latin-1")
        return (self.check_time, d(self.from_ip), d(self.from_type), d(self.from_uid), self.from_port,
                d(self.to_ip), d(self.to_type), d(self.to_uid), self.to_port)


SL_ERRORS = {1: "sock line is empty", 2: "closed socket on last entry", 3: "no smaller value found", 4: "closed socket"}
TCP_WIRE_SIZE = 64
TCP_ESTABLISHED, TCP_CLOSED = 1, 5


HPACK_EMIT = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)


class H2Out(C.Structure):
    _fields_ = [("method", C.c_char * 64), ("path", C.c_char * 1100), ("authority", C.c_char * OR_UID_MAX),
                ("protocol", C.c_char * 8), ("status_code", C.c_uint32), ("latency", C.c_uint64)]

    def as_tuple(self):
        return (self.method, self.path, self.authority, self.protocol, self.status_code, self.latency)


def _load():
    lib = C.CDLL(os.environ.get("SG_ORACLE_LIB_PATH") or build())   # override: sanitizer builds (tools/asan_host_tests.sh)
    P = C.c_void_p
    sig = {
        "or_create": (P, []), "or_destroy": (None, [P]),
        "or_set_clock": (None, [P, C.c_uint64, C.c_uint64]), "or_set_log_limit": (None, [P, C.c_size_t]),
        "or_process_pod": (C.c_int, [P, C.c_char_p, C.c_char_p, C.c_char_p]),
        "or_process_svc": (C.c_int, [P, C.c_char_p, C.c_char_p, C.c_char_p]),
        "or_process_l7_wire": (C.c_size_t, [P, C.c_void_p, C.c_size_t, C.c_void_p]),
        "or_process_packed": (C.c_size_t, [P, C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t]),
        "or_reqinfo_count": (C.c_size_t, [P]), "or_reqinfo_logged": (C.c_size_t, [P]),
        "or_reqinfo_at": (C.POINTER(ReqInfo), [P, C.c_size_t]),
        "or_dropped_src": (C.c_uint64, [P]), "or_dropped_parse": (C.c_uint64, [P]),
        "or_label_count": (C.c_size_t, [P]), "or_label_at": (C.c_char_p, [P, C.c_size_t]), "or_known_count": (C.c_size_t, [P]),
        "or_window_close": (C.c_size_t, [P, C.c_void_p, C.c_uint32]),
        "or_edge_count": (C.c_size_t, [P]), "or_edge_at": (C.POINTER(OrEdge), [P, C.c_size_t]),
        "or_node_count": (C.c_size_t, [P]),
        "or_node_features": (C.POINTER(C.c_float), [P]), "or_layer_output": (C.POINTER(C.c_float), [P, C.c_uint32]),
        "or_node_stats_sum": (C.POINTER(C.c_uint64), [P]), "or_node_stats_max": (C.POINTER(C.c_uint64), [P]),
        "or_outbound_ips": (C.POINTER(C.c_uint32), [P, C.POINTER(C.c_size_t)]),
        "or_window_tmin": (C.c_int64, [P]), "or_window_tmax": (C.c_int64, [P]), "or_window_events": (C.c_uint64, [P]),
        "or_parse_http_payload": (None, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
        "or_parse_postgres": (C.c_int, [P, C.c_uint32, C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
        "or_int_to_ipv4": (None, [C.c_uint32, C.c_char_p]),
        "or_weights_count": (C.c_size_t, [C.c_uint32]), "or_hash32": (C.c_uint32, [C.c_uint32]),
        "or_sl_create": (P, [C.c_uint32, C.c_uint64]), "or_sl_destroy": (None, [P]),
        "or_sl_add": (None, [P, C.c_uint64, C.POINTER(SockInfo)]),
        "or_sl_get": (C.c_int, [P, C.c_uint64, C.c_uint64, C.POINTER(SockInfo)]),
        "or_sl_delete_unused": (None, [P]), "or_sl_len": (C.c_size_t, [P]),
        "or_sl_at": (C.c_int, [P, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(SockInfo)]),
        "or_sl_seed_from_proc": (C.c_int, [P, C.c_char_p, C.c_uint64]),
        "or_sl_inode_from_link": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
        "or_sl_parse_tcp_line": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_int)]),
        "or_set_proc_root": (None, [P, C.c_char_p, C.c_uint64]), "or_process_exit": (None, [P, C.c_uint32]),
        "or_process_tcp": (C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint16, C.c_char_p, C.c_uint16]),
        "or_process_tcp_wire": (C.c_size_t, [P, C.c_void_p, C.c_size_t]),
        "or_sockline_of": (P, [P, C.c_uint32, C.c_uint64]), "or_sockline_count": (C.c_size_t, [P]), "or_pg_stmt_count": (C.c_size_t, [P]),
        "or_sweep_socket_lines": (C.c_size_t, [P, C.c_int64, C.c_int]),
        "or_alive_count": (C.c_size_t, [P]), "or_alive_at": (C.POINTER(Alive), [P, C.c_size_t]),
        "or_hpack_create": (P, [C.c_uint32]), "or_hpack_destroy": (None, [P]),
        "or_hpack_set_emit": (None, [P, HPACK_EMIT, C.c_void_p]),
        "or_hpack_write": (C.c_int, [P, C.c_char_p, C.c_size_t]),
        "or_hpack_dyn_len": (C.c_size_t, [P]), "or_hpack_dyn_size": (C.c_uint32, [P]),
        "or_hpack_dyn_at": (C.c_int, [P, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
        "or_hpack_selfcheck": (C.c_int, []), "or_hpack_huff_code": (C.c_uint32, [C.c_int, C.POINTER(C.c_uint8)]),
        "or_hpack_huff_decode": (C.c_long, [C.c_char_p, C.c_size_t, C.c_char_p]),
        "or_go_atoi_u32": (C.c_uint32, [C.c_char_p, C.c_size_t]),
        "or_h2_create": (P, []), "or_h2_destroy": (None, [P]),
        "or_h2_event": (C.c_int, [P, C.c_uint32, C.c_uint64, C.c_int, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(H2Out)]),
        "or_h2_proc_exec": (None, [P, C.c_uint32]), "or_h2_proc_exit": (None, [P, C.c_uint32]),
        "or_h2_conn_closed": (None, [P, C.c_uint32, C.c_uint64]), "or_h2_sweep": (None, [P]),
        "or_h2_pending": (C.c_size_t, [P]), "or_h2_parsers": (C.c_size_t, [P]),
        "or_h2_dropped_not_live": (C.c_uint64, [P]), "or_h2_dropped_unparsed": (C.c_uint64, [P]),
        "or_h2_of": (P, [P]),
        "or_set_kafka_decode": (None, [P, C.c_int]),
        "or_kafka_decode": (P, [C.c_char_p, C.c_size_t, C.c_int, C.c_int16]), "or_kafka_result_free": (None, [P]),
        "or_kafka_count": (C.c_size_t, [P]), "or_kafka_status": (C.c_int, [P]),
        "or_kafka_msg": (C.c_int, [P, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
        "or_kafka_decompress": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
        "or_kafka_free": (None, [P]),
        "or_crc32": (C.c_uint32, [C.c_int, C.c_char_p, C.c_size_t]), "or_xxh32": (C.c_uint32, [C.c_char_p, C.c_size_t, C.c_uint32]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name); f.restype = res; f.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def inode_from_link(link: str) -> Optional[str]:
    """getInodeFromFD's regexp (sock_num_line.go:358-363) on a link text."""
    buf = C.create_string_buffer(32)
    return buf.value.decode() if lib().or_sl_inode_from_link(link.encode(), buf, 32) == 0 else None


def parse_tcp_line(line: str):
    """parseTcpLine (sock_num_line.go:384-397) -> (local ip, local port, remote ip, remote port) or None."""
    a, b, lp, rp = C.create_string_buffer(16), C.create_string_buffer(16), C.c_int(), C.c_int()
    if lib().or_sl_parse_tcp_line(line.encode(), a, C.byref(lp), b, C.byref(rp)) != 0:
        return None
    return a.value.decode(), lp.value, b.value.decode(), rp.value


def sockinfo(saddr="", sport=0, daddr="", dport=0, pid=0, fd=0) -> SockInfo:
    return SockInfo(pid, fd, saddr.encode(), sport, daddr.encode(), dport)


class SockLine:
    """SocketLine (aggregator/sock_num_line.go).  Owns its C object unless wrapped around a borrowed pointer."""
    def __init__(self, pid: int = 0, fd: int = 0, _borrowed=None):
        self._l = lib()
        self._own = _borrowed is None
        self._s = self._l.or_sl_create(pid, fd) if self._own else _borrowed

    def __del__(self):
        try:
            if self._own and self._s:
                self._l.or_sl_destroy(self._s); self._s = None
        except Exception:
            pass

    def add(self, ts: int, si: Optional[SockInfo]):
        self._l.or_sl_add(self._s, ts, C.byref(si) if si is not None else None)

    def get(self, ts: int, now_ns: int = 1):
        """-> (SockInfo, None) or (None, error string), as GetValue returns (value, error)."""
        out = SockInfo()
        rc = self._l.or_sl_get(self._s, ts, now_ns, C.byref(out))
        return (out, None) if rc == 0 else (None, SL_ERRORS[rc])

    def delete_unused(self): self._l.or_sl_delete_unused(self._s)

    def seed_from_proc(self, proc_root: str, now_kernel_ns: int) -> int:
        """getConnectionInfo (sock_num_line.go:399-429) against `proc_root` instead of /proc; 0 = seeded."""
        return self._l.or_sl_seed_from_proc(self._s, proc_root.encode(), now_kernel_ns)

    def __len__(self): return self._l.or_sl_len(self._s)

    def values(self):
        """[(timestamp, last_match, (saddr, sport, daddr, dport) or None)]"""
        out = []
        for i in range(len(self)):
            ts, lm, si = C.c_uint64(), C.c_uint64(), SockInfo()
            o = self._l.or_sl_at(self._s, i, C.byref(ts), C.byref(lm), C.byref(si))
            out.append((ts.value, lm.value, (si.saddr.decode(), si.sport, si.daddr.decode(), si.dport) if o == 1 else None))
        return out


class Hpack:
    """hpack.Decoder of golang.org/x/net as oracle/http2.c restates it."""

    def __init__(self, max_table_size: int = 4096):
        self._l = lib(); self._d = self._l.or_hpack_create(max_table_size)
        self.fields: List[tuple] = []
        self._cb = HPACK_EMIT(lambda ctx, n, nl, v, vl: self.fields.append((C.string_at(n, nl), C.string_at(v, vl))))
        self._l.or_hpack_set_emit(self._d, self._cb, None)

    def __del__(self):
        try: self._l.or_hpack_destroy(self._d)
        except Exception: pass

    def write(self, block: bytes):
        """-> (rc, fields emitted by this call)"""
        self.fields = []
        rc = self._l.or_hpack_write(self._d, block, len(block))
        return rc, list(self.fields)

    def table(self):
        out = []
        for i in range(self._l.or_hpack_dyn_len(self._d)):
            n = C.c_void_p(); v = C.c_void_p(); nl = C.c_size_t(); vl = C.c_size_t()
            self._l.or_hpack_dyn_at(self._d, i, C.byref(n), C.byref(nl), C.byref(v), C.byref(vl))
            out.append((C.string_at(n, nl.value), C.string_at(v, vl.value)))
        return out

    def table_size(self) -> int: return self._l.or_hpack_dyn_size(self._d)


def huff_code(sym: int):
    ln = C.c_uint8(); code = lib().or_hpack_huff_code(sym, C.byref(ln)); return code, ln.value


def huff_encode(s: bytes) -> bytes:
    acc = 0; nb = 0
    for ch in s:
        code, ln = huff_code(ch); acc = (acc << ln) | code; nb += ln
    pad = (-nb) % 8
    acc = (acc << pad) | ((1 << pad) - 1); nb += pad
    return acc.to_bytes(nb // 8, "big") if nb else b""


def huff_decode(b: bytes):
    out = C.create_string_buffer(2 * len(b) + 1)
    r = lib().or_hpack_huff_decode(b, len(b), out)
    return None if r < 0 else out.raw[:r]


def go_atoi_u32(b: bytes) -> int: return lib().or_go_atoi_u32(b, len(b))


class H2Assembler:
    """processHttp2Frames as oracle/http2.c restates it (stand-alone or the Oracle's own)."""

    def __init__(self, _borrowed=None):
        self._l = lib(); self._own = _borrowed is None
        self._h = self._l.or_h2_create() if self._own else _borrowed

    def __del__(self):
        try:
            if self._own: self._l.or_h2_destroy(self._h)
        except Exception: pass

    def event(self, pid, fd, method_id, payload: bytes, write_ns, tls=False):
        out = H2Out()
        r = self._l.or_h2_event(self._h, pid, fd, method_id, payload, len(payload), write_ns, int(tls), C.byref(out))
        return out.as_tuple() if r else None

    def proc_exec(self, pid): self._l.or_h2_proc_exec(self._h, pid)
    def proc_exit(self, pid): self._l.or_h2_proc_exit(self._h, pid)
    def conn_closed(self, pid, fd): self._l.or_h2_conn_closed(self._h, pid, fd)
    def sweep(self): self._l.or_h2_sweep(self._h)
    def pending(self): return self._l.or_h2_pending(self._h)
    def parsers(self): return self._l.or_h2_parsers(self._h)
    def dropped_not_live(self): return self._l.or_h2_dropped_not_live(self._h)
    def dropped_unparsed(self): return self._l.or_h2_dropped_unparsed(self._h)


KAFKA_STATUS = {0: "ok", 1: "insufficient", 2: "error", 3: "panic"}


def kafka_decode(payload: bytes, method_id: int, api_version: int = 0):
    """decodeKafkaPayload as oracle/kafka.c restates it -> (status, [(topic, partition, key, value), ...])."""
    l = lib(); r = l.or_kafka_decode(payload, len(payload), method_id, api_version)
    try:
        out = []
        for i in range(l.or_kafka_count(r)):
            t = C.c_void_p(); k = C.c_void_p(); v = C.c_void_p(); tn = C.c_size_t(); kn = C.c_size_t(); vn = C.c_size_t(); part = C.c_int32()
            l.or_kafka_msg(r, i, C.byref(t), C.byref(tn), C.byref(part), C.byref(k), C.byref(kn), C.byref(v), C.byref(vn))
            out.append((C.string_at(t, tn.value), part.value, C.string_at(k, kn.value), C.string_at(v, vn.value)))
        return KAFKA_STATUS[l.or_kafka_status(r)], out
    finally:
        l.or_kafka_result_free(r)


def kafka_decompress(codec: int, data: bytes):
    l = lib(); out = C.c_void_p(); n = C.c_size_t()
    if l.or_kafka_decompress(codec, data, len(data), C.byref(out), C.byref(n)) != 0:
        return None
    try:
        return C.string_at(out, n.value)
    finally:
        l.or_kafka_free(out)


def crc32(data: bytes, castagnoli: bool = False) -> int: return lib().or_crc32(int(castagnoli), data, len(data))
def xxh32(data: bytes, seed: int = 0) -> int: return lib().or_xxh32(data, len(data), seed)


class Oracle:
    def __init__(self, first_kernel_ns: int = 0, first_user_ns: int = 0, log_limit: int = 0):
        self._l = lib()
        self._o = self._l.or_create()
        self._l.or_set_clock(self._o, first_kernel_ns, first_user_ns)
        self._l.or_set_log_limit(self._o, log_limit)

    def close(self):
        if self._o:
            self._l.or_destroy(self._o); self._o = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- k8s ---
    def pod(self, event_type: str, uid: str, ip: str) -> int:
        return self._l.or_process_pod(self._o, event_type.encode(), uid.encode(), ip.encode())

    def svc(self, event_type: str, uid: str, ip: str) -> int:
        return self._l.or_process_svc(self._o, event_type.encode(), uid.encode(), ip.encode())

    def apply_ops(self, ops):
        for kind, et, uid, ip in ops:
            (self.pod if kind == "pod" else self.svc)(et, uid, ip)

    # --- events ---
    def l7_wire(self, recs: bytes, kafka_msgs: Optional[np.ndarray] = None) -> int:
        n = len(recs) // L7_WIRE_SIZE
        km = None
        if kafka_msgs is not None:
            kafka_msgs = np.ascontiguousarray(kafka_msgs, dtype=np.uint32); km = kafka_msgs.ctypes.data
        if not isinstance(recs, bytes):
            recs = bytes(recs)
        # no copy: the C side only reads (a 219 MB from_buffer_copy per call was a third of the measured "CPU baseline")
        return self._l.or_process_l7_wire(self._o, C.cast(C.c_char_p(recs), C.c_void_p), n, km)

    def packed(self, events: np.ndarray, labels: Sequence[str]) -> int:
        ev = np.ascontiguousarray(events)
        arr = (C.c_char_p * max(1, len(labels)))(*[s.encode() for s in labels])
        return self._l.or_process_packed(self._o, ev.ctypes.data, len(ev), arr, len(labels))

    # --- results ---
    # ---- f-2: TCP connect events -> socket lines -> alive connections ----
    def tcp(self, type_: int, pid: int, fd: int, ts: int, saddr: str, sport: int, daddr: str, dport: int) -> int:
        return self._l.or_process_tcp(self._o, type_, pid, fd, ts, saddr.encode(), sport, daddr.encode(), dport)

    def tcp_wire(self, recs: bytes) -> int:
        assert len(recs) % TCP_WIRE_SIZE == 0
        return self._l.or_process_tcp_wire(self._o, recs, len(recs) // TCP_WIRE_SIZE)

    def sockline(self, pid: int, fd: int) -> Optional[SockLine]:
        p = self._l.or_sockline_of(self._o, pid, fd)
        return SockLine(_borrowed=p) if p else None

    def sockline_count(self) -> int: return self._l.or_sockline_count(self._o)

    def set_proc_root(self, root: Optional[str], now_user_ns: int = 0):
        self._l.or_set_proc_root(self._o, root.encode() if root is not None else None, now_user_ns)

    def process_exit(self, pid: int): self._l.or_process_exit(self._o, pid)

    def pg_stmt_count(self) -> int: return self._l.or_pg_stmt_count(self._o)

    def set_kafka_decode(self, on: bool = True): self._l.or_set_kafka_decode(self._o, int(on))

    def h2(self) -> H2Assembler: return H2Assembler(_borrowed=self._l.or_h2_of(self._o))

    def sweep(self, now_ms: int, send_alive: bool = True) -> int:
        return self._l.or_sweep_socket_lines(self._o, now_ms, int(send_alive))

    def alive_count(self) -> int: return self._l.or_alive_count(self._o)

    def alive_rows(self):
        out, i = [], 0
        while True:
            p = self._l.or_alive_at(self._o, i)
            if not p:
                return out
            out.append(p.contents.as_tuple()); i += 1

    def window_close(self, weights: np.ndarray, layers: int) -> int:
        w = np.ascontiguousarray(weights, dtype=np.float32)
        assert len(w) == self._l.or_weights_count(layers)
        return self._l.or_window_close(self._o, w.ctypes.data, layers)

    def edges(self) -> List[OrEdge]:
        n = self._l.or_edge_count(self._o)
        return [self._l.or_edge_at(self._o, i).contents for i in range(n)]

    def edge_rows(self) -> np.ndarray:
        """Closed-window edges as a numpy array with the sg_edge_out layout."""
        from alaz_amd.replay import EDGE_OUT_DTYPE
        n = self._l.or_edge_count(self._o)
        out = np.zeros(n, dtype=EDGE_OUT_DTYPE)
        for i in range(n):
            e = self._l.or_edge_at(self._o, i).contents
            out[i] = np.frombuffer(bytes(e.row), dtype=EDGE_OUT_DTYPE)[0]
        return out

    def edge_hist(self) -> np.ndarray:
        """[edges][16] u32 latency histogram bins of the closed window, row order (f-3)."""
        n = self._l.or_edge_count(self._o)
        self._l.or_edge_hist.restype = C.POINTER(C.c_uint32); self._l.or_edge_hist.argtypes = [C.c_void_p]
        p = self._l.or_edge_hist(self._o)
        return np.ctypeslib.as_array(p, shape=(n, 16)).copy() if n else np.zeros((0, 16), np.uint32)

    def edge_dict(self):
        """{(from_type, from_uid, to_type, to_uid): (count, err, sum, max, sumsq, score, lat_z, err_ratio, alive, p50_us, p99_us)}"""
        d = {}
        for e in self.edges():
            k = (e.from_type.decode(), e.from_uid.decode(), e.to_type.decode(), e.to_uid.decode())
            r = e.row
            d[k] = (r.count, r.err_count, r.sum_ns, r.max_ns, r.sumsq_us, r.score, r.lat_z, r.err_ratio, r.alive, r.p50_us, r.p99_us)
        return d

    def reqinfos(self):
        n = self._l.or_reqinfo_logged(self._o)
        return [self._l.or_reqinfo_at(self._o, i).contents.as_tuple() for i in range(n)]

    @property
    def persisted(self): return self._l.or_reqinfo_count(self._o)
    @property
    def dropped_src(self): return self._l.or_dropped_src(self._o)
    @property
    def dropped_parse(self): return self._l.or_dropped_parse(self._o)
    @property
    def n_nodes(self): return self._l.or_node_count(self._o)
    @property
    def n_known(self): return self._l.or_known_count(self._o)
    @property
    def labels(self): return [self._l.or_label_at(self._o, i).decode() for i in range(self._l.or_label_count(self._o))]
    @property
    def window_tmin(self): return self._l.or_window_tmin(self._o)
    @property
    def window_tmax(self): return self._l.or_window_tmax(self._o)
    @property
    def window_events(self): return self._l.or_window_events(self._o)

    def node_features(self) -> np.ndarray:
        n = self.n_nodes
        return np.ctypeslib.as_array(self._l.or_node_features(self._o), shape=(n, 32)).copy()

    def layer_output(self, l: int) -> np.ndarray:
        n = self.n_nodes
        return np.ctypeslib.as_array(self._l.or_layer_output(self._o, l), shape=(n, 64)).copy()

    def node_stats(self):
        n = self.n_nodes
        s = np.ctypeslib.as_array(self._l.or_node_stats_sum(self._o), shape=(n, 10)).copy()
        m = np.ctypeslib.as_array(self._l.or_node_stats_max(self._o), shape=(n, 2)).copy()
        return s, m

    def outbound_ips(self) -> np.ndarray:
        k = C.c_size_t(0)
        p = self._l.or_outbound_ips(self._o, C.byref(k))
        return np.ctypeslib.as_array(p, shape=(k.value,)).copy() if k.value else np.zeros(0, np.uint32)

    # --- stand-alone pieces ---
    def parse_postgres(self, pid: int, fd: int, method: str, payload: bytes):
        out = C.create_string_buffer(2048)
        rc = self._l.or_parse_postgres(self._o, pid, fd, method.encode(), payload, len(payload), out, 2048)
        return rc, out.value.decode("latin-1")


def parse_http_payload(req: bytes):
    m = C.create_string_buffer(64); p = C.create_string_buffer(1100); v = C.create_string_buffer(64); h = C.create_string_buffer(OR_UID_MAX)
    lib().or_parse_http_payload(req, len(req), m, p, v, h)
    return m.value.decode("latin-1"), p.value.decode("latin-1"), v.value.decode("latin-1"), h.value.decode("latin-1")


def int_to_ipv4(ip: int) -> str:
    b = C.create_string_buffer(16); lib().or_int_to_ipv4(ip, b); return b.value.decode()


# ------------------------------------------------------------------------------------------------
# lean CPU baseline (oracle/lean_baseline.c): the same join + per-edge aggregation a careful CPU
# implementation would do on u32 keys.  Measurement infrastructure, like the rest of oracle/.
# ------------------------------------------------------------------------------------------------
class Lean:
    def __init__(self, max_ips: int, max_edges: int, max_labels: int = 1 << 20):
        l = lib()
        l.lean_create.restype = C.c_void_p; l.lean_create.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32]
        l.lean_destroy.argtypes = [C.c_void_p]
        l.lean_upsert.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        l.lean_reset_window.argtypes = [C.c_void_p]
        l.lean_process.restype = C.c_size_t; l.lean_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for f in ("lean_edges", "lean_accepted", "lean_dropped_src"):
            getattr(l, f).restype = C.c_uint64; getattr(l, f).argtypes = [C.c_void_p]
        l.lean_checksums.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self._l, self._h = l, l.lean_create(max_ips, max_edges, max_labels)

    def close(self):
        if self._h:
            self._l.lean_destroy(self._h); self._h = None

    def upsert_pod(self, ip: int, node_id: int): self._l.lean_upsert(self._h, ip, 1, node_id)
    def upsert_service(self, ip: int, node_id: int): self._l.lean_upsert(self._h, ip, 2, node_id)
    def reset_window(self): self._l.lean_reset_window(self._h)

    def process(self, events: np.ndarray) -> int:
        ev = np.ascontiguousarray(events)
        return self._l.lean_process(self._h, ev.ctypes.data, len(ev))        # ctypes releases the GIL for the call

    @property
    def edges(self): return self._l.lean_edges(self._h)
    @property
    def accepted(self): return self._l.lean_accepted(self._h)
    @property
    def dropped_src(self): return self._l.lean_dropped_src(self._h)

    def checksums(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._l.lean_checksums(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
