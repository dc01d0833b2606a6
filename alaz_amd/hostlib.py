"""ctypes binding of libsgdatastore.so — the C++ host side (DataStore mirror, L7 packer, GraphDS).
Plumbing only; see alaz_amd/csrc/host/."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from .engine import LIB_PATH as ENGINE_LIB, SgConfig
from .replay import EVENT_DTYPE, L7_WIRE_SIZE

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(_HERE, "lib", "libsgdatastore.so")


class EdgeRowC(C.Structure):
    _fields_ = [("from_type", C.c_char * 12), ("to_type", C.c_char * 12), ("from_uid", C.c_char * 160), ("to_uid", C.c_char * 160),
                ("count", C.c_uint32), ("err_count", C.c_uint32), ("sum_ns", C.c_uint64), ("max_ns", C.c_uint64), ("sumsq_us", C.c_uint64),
                ("score", C.c_float), ("lat_z", C.c_float), ("err_ratio", C.c_float), ("alive", C.c_uint32),
                ("p50_us", C.c_uint32), ("p99_us", C.c_uint32)]


class SockInfoC(C.Structure):
    _fields_ = [("pid", C.c_uint32), ("fd", C.c_uint64), ("saddr", C.c_uint32), ("sport", C.c_uint16), ("daddr", C.c_uint32), ("dport", C.c_uint16)]


_lib = None


class H2OutC(C.Structure):
    _fields_ = [("method", C.c_char * 64), ("path", C.c_char * 1100), ("authority", C.c_char * 160), ("protocol", C.c_char * 8),
                ("status_code", C.c_uint32), ("latency", C.c_uint64)]

    def as_tuple(self):
        return (self.method, self.path, self.authority, self.protocol, self.status_code, self.latency)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("SG_HOST_LIB_PATH") or HOST_LIB       # override: sanitizer builds (tools/asan_host_tests.sh)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m alaz_amd.build`")
        lib = C.CDLL(path)
        P, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
        sig = {
            "sgh_packer_create": (P, []), "sgh_packer_destroy": (None, [P]), "sgh_packer_known_ip": (None, [P, u32, C.c_int]),
            "sgh_packer_pack_wire": (sz, [P, P, sz, P, P, sz]), "sgh_packer_pack_wire_full": (sz, [P, P, sz, P, P, sz]), "sgh_packer_labels": (sz, [P, C.c_char_p, sz]),
            "sgh_packer_dropped_parse": (u64, [P]),
            "sgh_parse_http": (None, [C.c_char_p, sz, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz]),
            "sgh_graphds_create": (P, [C.c_char_p, C.POINTER(SgConfig), sz]), "sgh_graphds_destroy": (None, [P]),
            "sgh_graphds_persist_pod": (C.c_int, [P, C.c_char_p, C.c_char_p, C.c_char_p]),
            "sgh_graphds_persist_service": (C.c_int, [P, C.c_char_p, C.c_char_p, C.c_char_p]),
            "sgh_graphds_ingest_wire": (C.c_int, [P, P, sz, P]),
            "sgh_graphds_persist_request": (C.c_int, [P, C.c_int64, u64, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                                      C.c_char_p, u32, C.c_char_p, C.c_int]),
            "sgh_graphds_flush": (C.c_long, [P, C.c_int64, C.POINTER(EdgeRowC), sz]),
            "sgh_sockline_create": (P, [u32, u64]), "sgh_sockline_destroy": (None, [P]),
            "sgh_sockline_add": (None, [P, u64, C.POINTER(SockInfoC)]), "sgh_sockline_get": (C.c_int, [P, u64, u64, C.POINTER(SockInfoC)]),
            "sgh_sockline_seed": (C.c_int, [P, C.c_char_p, u64]),
            "sgh_proc_inode_of_link": (C.c_int, [C.c_char_p, C.c_char_p, sz]),
            "sgh_proc_parse_tcp_line": (C.c_int, [C.c_char_p, C.POINTER(u32), C.POINTER(C.c_uint16), C.POINTER(u32), C.POINTER(C.c_uint16)]),
            "sgh_graphds_pg_statements": (sz, [P]),
            "sgh_graphds_set_proc_root": (None, [P, C.c_char_p, u64, u64, u64]), "sgh_graphds_seed_stats": (None, [P, C.POINTER(u64)]),
            "sgh_sockline_delete_unused": (None, [P]), "sgh_sockline_len": (sz, [P]),
            "sgh_sockline_at": (C.c_int, [P, sz, C.POINTER(u64), C.POINTER(u64), C.POINTER(SockInfoC)]),
            "sgh_graphds_tcp_wire": (sz, [P, P, sz]), "sgh_graphds_socklines": (sz, [P]), "sgh_graphds_sockline": (P, [P, u32, u64]),
            "sgh_sockline_ref_get": (P, [P]), "sgh_sockline_ref_release": (None, [P]),
            "sgh_graphds_sweep": (sz, [P, C.c_int64, C.c_int]),
            "sgh_graphds_labels": (sz, [P, C.c_char_p, sz]), "sgh_graphds_dropped_parse": (u64, [P]), "sgh_graphds_engine": (P, [P]),
            "sgh_graphds_proc_exec": (None, [P, u32]), "sgh_graphds_proc_exit": (None, [P, u32]), "sgh_graphds_sweep_http2": (None, [P]),
            "sgh_graphds_http2_stats": (None, [P, C.POINTER(u64)]),
            "sgh_hpack_create": (P, [u32]), "sgh_hpack_destroy": (None, [P]),
            "sgh_hpack_write": (C.c_long, [P, C.c_char_p, sz, C.c_char_p, sz, C.POINTER(u32), sz]),
            "sgh_hpack_table_len": (sz, [P]), "sgh_hpack_table_size": (u32, [P]),
            "sgh_hpack_table_at": (sz, [P, sz, C.c_char_p, sz, C.POINTER(u32)]),
            "sgh_huffman_decode": (C.c_long, [C.c_char_p, sz, C.c_char_p, sz]), "sgh_huffman_encode": (C.c_long, [C.c_char_p, sz, C.c_char_p, sz]),
            "sgh_go_atoi_u32": (u32, [C.c_char_p, sz]),
            "sgh_h2_create": (P, []), "sgh_h2_destroy": (None, [P]),
            "sgh_h2_event": (C.c_int, [P, u32, u64, C.c_int, C.c_char_p, u32, u64, C.c_int, C.POINTER(H2OutC)]),
            "sgh_h2_proc_exec": (None, [P, u32]), "sgh_h2_proc_exit": (None, [P, u32]), "sgh_h2_conn_closed": (None, [P, u32, u64]),
            "sgh_h2_sweep": (None, [P]), "sgh_h2_pending": (sz, [P]), "sgh_h2_parsers": (sz, [P]),
            "sgh_packer_proc_exec": (None, [P, u32]), "sgh_packer_proc_exit": (None, [P, u32]), "sgh_packer_conn_closed": (None, [P, u32, u64]), "sgh_packer_pg_statements": (sz, [P]),
            "sgh_graphds_edges_json": (C.c_long, [P, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz, C.c_char_p, sz]),
            "sgh_json_string": (sz, [C.c_char_p, sz, C.c_char_p, sz]),
            "sgh_edges_json_from_rows": (C.c_long, [C.POINTER(EdgeRowC), sz, C.c_int64, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz, C.c_char_p, sz]),
            "sgh_packer_kafka_decode": (None, [P, C.c_int]), "sgh_graphds_kafka_decode": (None, [P, C.c_int]),
            "sgh_kafka_decode": (C.c_long, [C.c_char_p, sz, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_char_p, sz]),
            "sgh_kafka_decompress": (C.c_long, [C.c_int, C.c_char_p, sz, C.c_char_p, sz]),
            "sgh_crc32": (u32, [C.c_int, C.c_char_p, sz]), "sgh_xxh32": (u32, [C.c_char_p, sz, u32]),
            "sgh_graphds_create2": (P, [C.c_char_p, P, sz, C.c_int]), "sgh_graphds_counters": (None, [P, P]),
            "sgh_mock_events": (sz, [P, P, sz]), "sgh_mock_table_ops": (sz, [P, P, sz]), "sgh_mock_label_count": (u32, [P]),
        }
        for name, (res, args) in sig.items():
            f = getattr(lib, name); f.restype = res; f.argtypes = args
        _lib = lib
    return _lib


def _labels(fn, h) -> List[str]:
    buf = C.create_string_buffer(1 << 16)
    n = fn(h, buf, len(buf))
    return buf.value.decode().split("\n") if n else []


def parse_http(req: bytes):
    m, p, v, h = (C.create_string_buffer(1200) for _ in range(4))
    load().sgh_parse_http(req, len(req), m, p, v, h, 1200)
    return tuple(x.value.decode("latin-1") for x in (m, p, v, h))


def proc_inode_of_link(link: str):
    buf = C.create_string_buffer(32)
    return buf.value.decode() if load().sgh_proc_inode_of_link(link.encode(), buf, 32) == 0 else None


def proc_parse_tcp_line(line: str):
    """-> (local addr u32, local port, remote addr u32, remote port) or None"""
    la, ra, lp, rp = C.c_uint32(), C.c_uint32(), C.c_uint16(), C.c_uint16()
    if load().sgh_proc_parse_tcp_line(line.encode(), C.byref(la), C.byref(lp), C.byref(ra), C.byref(rp)) != 0:
        return None
    return la.value, lp.value, ra.value, rp.value


SL_ERRORS = {1: "sock line is empty", 2: "closed socket on last entry", 3: "no smaller value found", 4: "closed socket"}


class SocketLine:
    """alaz::SocketLine (csrc/host/sockline.hpp); addresses are numeric IPv4."""
    def __init__(self, pid: int = 0, fd: int = 0, _ref=None):
        # _ref: an owning reference handed out by sgh_graphds_sockline (a line of a GraphDS's tracker: it stays alive here after the
        # tracker dropped it at process exit)
        self._l = load(); self._own = _ref is None; self._ref = _ref
        self._p = self._l.sgh_sockline_create(pid, fd) if self._own else self._l.sgh_sockline_ref_get(_ref)

    def __del__(self):
        try:
            if self._own and self._p:
                self._l.sgh_sockline_destroy(self._p); self._p = None
            elif self._ref:
                self._l.sgh_sockline_ref_release(self._ref); self._ref = None; self._p = None
        except Exception:
            pass

    def add(self, ts: int, si):
        """si = (saddr, sport, daddr, dport) or None for a close"""
        if si is None:
            self._l.sgh_sockline_add(self._p, ts, None)
        else:
            c = SockInfoC(0, 0, si[0], si[1], si[2], si[3]); self._l.sgh_sockline_add(self._p, ts, C.byref(c))

    def get(self, ts: int, now_ns: int = 1):
        out = SockInfoC()
        rc = self._l.sgh_sockline_get(self._p, ts, now_ns, C.byref(out))
        return ((out.saddr, out.sport, out.daddr, out.dport), None) if rc == 0 else (None, SL_ERRORS[rc])

    def delete_unused(self): self._l.sgh_sockline_delete_unused(self._p)

    def seed_from_proc(self, proc_root: str, now_kernel_ns: int) -> int:
        """getConnectionInfo against `proc_root` (aggregator/sock_num_line.go:399-429); 0 = seeded"""
        return self._l.sgh_sockline_seed(self._p, proc_root.encode(), now_kernel_ns)

    def __len__(self): return self._l.sgh_sockline_len(self._p)

    def values(self):
        out = []
        for i in range(len(self)):
            ts, lm, si = C.c_uint64(), C.c_uint64(), SockInfoC()
            o = self._l.sgh_sockline_at(self._p, i, C.byref(ts), C.byref(lm), C.byref(si))
            out.append((ts.value, lm.value, (si.saddr, si.sport, si.daddr, si.dport) if o == 1 else None))
        return out


class Hpack:
    """hpack::Decoder (csrc/host/http2.hpp)."""

    def __init__(self, max_table_size: int = 4096):
        self._l = load(); self._d = self._l.sgh_hpack_create(max_table_size)

    def __del__(self):
        try: self._l.sgh_hpack_destroy(self._d)
        except Exception: pass

    def write(self, block: bytes):
        """-> (rc, fields): rc 0 ok / -1 decoding error; fields emitted by this call"""
        cap = 8 * len(block) + 4096; buf = C.create_string_buffer(cap); lens = (C.c_uint32 * 512)()
        r = self._l.sgh_hpack_write(self._d, block, len(block), buf, cap, lens, 256)
        nf = r if r >= 0 else -r - 1
        out = []; off = 0
        for i in range(nf):
            nl, vl = lens[2 * i], lens[2 * i + 1]
            out.append((buf.raw[off:off + nl], buf.raw[off + nl:off + nl + vl])); off += nl + vl
        return (0 if r >= 0 else -1), out

    def table(self):
        out = []
        for i in range(self._l.sgh_hpack_table_len(self._d)):
            buf = C.create_string_buffer(8192); lens = (C.c_uint32 * 2)()
            self._l.sgh_hpack_table_at(self._d, i, buf, 8192, lens)
            out.append((buf.raw[:lens[0]], buf.raw[lens[0]:lens[0] + lens[1]]))
        return out

    def table_size(self) -> int: return self._l.sgh_hpack_table_size(self._d)


def huffman_decode(b: bytes):
    out = C.create_string_buffer(2 * len(b) + 8)
    r = load().sgh_huffman_decode(b, len(b), out, 2 * len(b) + 8)
    return None if r < 0 else out.raw[:r]


def huffman_encode(b: bytes) -> bytes:
    out = C.create_string_buffer(4 * len(b) + 8)
    r = load().sgh_huffman_encode(b, len(b), out, 4 * len(b) + 8)
    return out.raw[:r]


def go_atoi_u32(b: bytes) -> int: return load().sgh_go_atoi_u32(b, len(b))


def edges_json_from_rows(rows, window_end_ms=0, monitoring_id="", idempotency_key="", node_id="", version="", batch=1000) -> List[str]:
    """rows: [(from_type, from_uid, to_type, to_uid, count, err, sum_ns, max_ns, sumsq_us, score, lat_z, err_ratio, alive[, p50_us, p99_us])] with bytes strings"""
    arr = (EdgeRowC * max(1, len(rows)))()
    for a, r in zip(arr, rows):
        a.from_type, a.from_uid, a.to_type, a.to_uid = r[0], r[1], r[2], r[3]
        a.count, a.err_count, a.sum_ns, a.max_ns, a.sumsq_us, a.score, a.lat_z, a.err_ratio, a.alive = r[4:13]
        if len(r) > 14:
            a.p50_us, a.p99_us = r[13], r[14]
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        n = load().sgh_edges_json_from_rows(arr, len(rows), window_end_ms, monitoring_id.encode(), idempotency_key.encode(), node_id.encode(), version.encode(), batch, buf, cap)
        if n >= 0:
            return buf.value.decode("utf-8").split("\n") if n else []
        cap = -n + 16


def json_string(b: bytes) -> str:
    n = load().sgh_json_string(b, len(b), None, 0); out = C.create_string_buffer(n + 1)
    load().sgh_json_string(b, len(b), out, n + 1)
    return out.value.decode("utf-8")


class Http2Assembler:
    def __init__(self):
        self._l = load(); self._a = self._l.sgh_h2_create()

    def __del__(self):
        try: self._l.sgh_h2_destroy(self._a)
        except Exception: pass

    def event(self, pid, fd, method_id, payload: bytes, write_ns, tls=False):
        out = H2OutC()
        r = self._l.sgh_h2_event(self._a, pid, fd, method_id, payload, len(payload), write_ns, int(tls), C.byref(out))
        return out.as_tuple() if r else None

    def proc_exec(self, pid): self._l.sgh_h2_proc_exec(self._a, pid)
    def proc_exit(self, pid): self._l.sgh_h2_proc_exit(self._a, pid)
    def conn_closed(self, pid, fd): self._l.sgh_h2_conn_closed(self._a, pid, fd)
    def sweep(self): self._l.sgh_h2_sweep(self._a)
    def pending(self): return self._l.sgh_h2_pending(self._a)
    def parsers(self): return self._l.sgh_h2_parsers(self._a)


KAFKA_STATUS = {0: "ok", 1: "insufficient", 2: "error", 3: "panic"}


def kafka_decode(payload: bytes, method_id: int, api_version: int = 0):
    """kafka::DecodePayload -> (status, [(topic, partition, key, value), ...])"""
    import struct
    st = C.c_int(); cap = 1 << 20; buf = C.create_string_buffer(cap)
    n = load().sgh_kafka_decode(payload, len(payload), method_id, api_version, C.byref(st), buf, cap)
    out = []; off = 0; raw = buf.raw
    for _ in range(n):
        tn, part, kn, vn = struct.unpack_from("<IiII", raw, off); off += 16
        out.append((raw[off:off + tn], part, raw[off + tn:off + tn + kn], raw[off + tn + kn:off + tn + kn + vn])); off += tn + kn + vn
    return KAFKA_STATUS[st.value], out


def kafka_decompress(codec: int, data: bytes):
    cap = 1 << 22; out = C.create_string_buffer(cap)
    r = load().sgh_kafka_decompress(codec, data, len(data), out, cap)
    return None if r < 0 else out.raw[:r]


def crc32(data: bytes, castagnoli: bool = False) -> int: return load().sgh_crc32(int(castagnoli), data, len(data))
def xxh32(data: bytes, seed: int = 0) -> int: return load().sgh_xxh32(data, len(data), seed)


class Packer:
    def __init__(self):
        self._l = load(); self._p = self._l.sgh_packer_create()

    def __del__(self):
        try:
            self._l.sgh_packer_destroy(self._p)
        except Exception:
            pass

    def known_ip(self, ip: int, add: bool = True): self._l.sgh_packer_known_ip(self._p, ip, int(add))

    def pack_wire(self, wire: bytes, kafka_msgs: Optional[np.ndarray] = None, full_copy: bool = False) -> np.ndarray:
        n = len(wire) // L7_WIRE_SIZE
        km = None
        if kafka_msgs is not None:
            kafka_msgs = np.ascontiguousarray(kafka_msgs, dtype=np.uint32); km = kafka_msgs.ctypes.data
            cap = int(kafka_msgs.sum()) + n
        else:
            cap = n * 160 if getattr(self, "_kafka", False) else n      # a 1 KiB payload holds < 160 minimal records
        out = np.zeros(max(cap, 1), dtype=EVENT_DTYPE)
        fn = self._l.sgh_packer_pack_wire_full if full_copy else self._l.sgh_packer_pack_wire
        k = fn(self._p, C.cast(C.c_char_p(wire), C.c_void_p), n, km, out.ctypes.data, cap)
        return out[:k]

    @property
    def labels(self): return _labels(self._l.sgh_packer_labels, self._p)
    @property
    def dropped_parse(self): return self._l.sgh_packer_dropped_parse(self._p)

    def kafka_decode(self, on: bool = True): self._kafka = on; self._l.sgh_packer_kafka_decode(self._p, int(on))
    def proc_exec(self, pid): self._l.sgh_packer_proc_exec(self._p, pid)
    def proc_exit(self, pid): self._l.sgh_packer_proc_exit(self._p, pid)
    def conn_closed(self, pid, fd): self._l.sgh_packer_conn_closed(self._p, pid, fd)
    def pg_statements(self) -> int: return self._l.sgh_packer_pg_statements(self._p)


class GraphDS:
    """C++ GraphDS over the real engine (engine_lib = path of libservicegraph.so) or over a recording
    stand-in (engine_lib=None; host-logic tests)."""

    def __init__(self, cfg: SgConfig, engine_lib: Optional[str] = ENGINE_LIB, batch: int = 4096, divert_requests: bool = False):
        self._l = load()
        self._g = self._l.sgh_graphds_create2(engine_lib.encode() if engine_lib else None, C.byref(cfg), batch, int(divert_requests))
        if not self._g:
            raise RuntimeError("GraphDS: engine could not be created (no usable gfx950 device or library missing); no CPU fallback")
        self.max_edges = int(cfg.max_edges)

    def close(self):
        if self._g:
            self._l.sgh_graphds_destroy(self._g); self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def PersistPod(self, uid: str, ip: str, event_type: str = "ADD"): return self._l.sgh_graphds_persist_pod(self._g, event_type.encode(), uid.encode(), ip.encode())
    def PersistService(self, uid: str, ip: str, event_type: str = "ADD"): return self._l.sgh_graphds_persist_service(self._g, event_type.encode(), uid.encode(), ip.encode())

    def apply_ops(self, ops):
        for kind, et, uid, ip in ops:
            (self.PersistPod if kind == "pod" else self.PersistService)(uid, ip, et)

    def ingest_wire(self, wire: bytes, kafka_msgs: Optional[np.ndarray] = None) -> int:
        n = len(wire) // L7_WIRE_SIZE
        km = None
        if kafka_msgs is not None:
            kafka_msgs = np.ascontiguousarray(kafka_msgs, dtype=np.uint32); km = kafka_msgs.ctypes.data
        buf = (C.c_char * len(wire)).from_buffer_copy(wire)
        return self._l.sgh_graphds_ingest_wire(self._g, C.addressof(buf), n, km)

    def PersistRequest(self, r: Sequence) -> int:
        """r: the 16 ReqInfo slots (datastore/backend.go:824-839)."""
        e = lambda s: s.encode("latin-1")
        return self._l.sgh_graphds_persist_request(self._g, r[0], r[1], e(r[2]), e(r[3]), e(r[4]), e(r[6]), e(r[7]), e(r[8]), e(r[10]), r[11], e(r[13]), int(r[15]))

    def FlushWindow(self, window_end_ms: int = 0):
        out = (EdgeRowC * self.max_edges)()
        n = self._l.sgh_graphds_flush(self._g, window_end_ms, out, self.max_edges)
        if n < 0:
            raise RuntimeError(f"FlushWindow rc={n}")
        d = {}
        for i in range(min(n, self.max_edges)):
            r = out[i]
            d[(r.from_type.decode(), r.from_uid.decode(), r.to_type.decode(), r.to_uid.decode())] = (
                r.count, r.err_count, r.sum_ns, r.max_ns, r.sumsq_us, r.score, r.lat_z, r.err_ratio, r.alive, r.p50_us, r.p99_us)
        return d

    # ---- f-2: TCP connect events -> socket lines -> alive connections ----
    def tcp_wire(self, recs: bytes) -> int:
        assert len(recs) % 64 == 0
        buf = (C.c_uint8 * len(recs)).from_buffer_copy(recs)
        return self._l.sgh_graphds_tcp_wire(self._g, C.addressof(buf), len(recs) // 64)

    def sockline_count(self) -> int: return self._l.sgh_graphds_socklines(self._g)

    def sockline(self, pid: int, fd: int):
        p = self._l.sgh_graphds_sockline(self._g, pid, fd)
        return SocketLine(_ref=p) if p else None

    def sweep(self, now_ms: int, send_alive: bool = True) -> int: return self._l.sgh_graphds_sweep(self._g, now_ms, int(send_alive))

    def set_proc_root(self, root, first_kernel_ns: int = 0, first_user_ns: int = 0, now_user_ns: int = 0):
        """NewSocketLine(fetch = true): lines created from now on are seeded from `root`/<pid>/... (None = off)"""
        self._l.sgh_graphds_set_proc_root(self._g, root.encode() if root else None, first_kernel_ns, first_user_ns, now_user_ns)

    def pg_statements(self) -> int: return self._l.sgh_graphds_pg_statements(self._g)

    def seed_stats(self):
        out = (C.c_uint64 * 2)(); self._l.sgh_graphds_seed_stats(self._g, out); return out[0], out[1]

    def edges_json(self, monitoring_id="", idempotency_key="", node_id="", version="", batch=1000) -> List[str]:
        """The rows of the last FlushWindow as "/edges/" payloads (edges_payload.hpp), one JSON document per batch."""
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            n = self._l.sgh_graphds_edges_json(self._g, monitoring_id.encode(), idempotency_key.encode(), node_id.encode(), version.encode(), batch, buf, cap)
            if n >= 0:
                return buf.value.decode("utf-8").split("\n") if n else []
            cap = -n + 16

    def kafka_decode(self, on: bool = True): self._l.sgh_graphds_kafka_decode(self._g, int(on))
    def proc_exec(self, pid: int): self._l.sgh_graphds_proc_exec(self._g, pid)
    def proc_exit(self, pid: int): self._l.sgh_graphds_proc_exit(self._g, pid)
    def sweep_http2(self): self._l.sgh_graphds_sweep_http2(self._g)

    def http2_stats(self):
        out = (C.c_uint64 * 5)(); self._l.sgh_graphds_http2_stats(self._g, out)
        return dict(zip(("pending", "parsers", "dropped_not_live", "dropped_unparsed", "dropped_time"), list(out)))

    @property
    def labels(self): return _labels(self._l.sgh_graphds_labels, self._g)
    @property
    def dropped_parse(self): return self._l.sgh_graphds_dropped_parse(self._g)
    @property
    def engine_handle(self): return self._l.sgh_graphds_engine(self._g)

    # engine settings that are not part of the DataStore surface go straight to the C ABI
    def set_clock(self, first_kernel_ns: int, first_user_ns: int):
        from . import engine
        assert engine.load_library().sg_set_clock(self.engine_handle, first_kernel_ns, first_user_ns) == 0

    def load_weights(self, w: np.ndarray):
        from . import engine
        w = np.ascontiguousarray(w, dtype=np.float32)
        assert engine.load_library().sg_load_weights(self.engine_handle, w.ctypes.data, len(w)) == 0

    def counters(self) -> dict:
        a = (C.c_uint64 * 9)()
        self._l.sgh_graphds_counters(self._g, a)
        return dict(zip(("offered", "batches_dropped", "engine_errors", "live_ids", "inner_requests", "inner_kafka", "inner_alive", "inner_pods", "inner_services"), list(a)))

    def mock_events(self) -> np.ndarray:
        n = self._l.sgh_mock_events(self._g, None, 0)
        out = np.zeros(max(n, 1), dtype=EVENT_DTYPE)
        self._l.sgh_mock_events(self._g, out.ctypes.data, n)
        return out[:n]

    def mock_table_ops(self) -> np.ndarray:
        n = self._l.sgh_mock_table_ops(self._g, None, 0)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        self._l.sgh_mock_table_ops(self._g, out.ctypes.data, n)
        return out[:n]

    @property
    def mock_label_count(self): return self._l.sgh_mock_label_count(self._g)
