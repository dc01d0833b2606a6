"""HTTP/2 payload builders for the tests: an HPACK *encoder* (RFC 7541 §6) and frame headers (RFC 7540 §4.1).
Test infrastructure only; Huffman coding comes from the oracle's table (oracle/http2.c)."""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

from oracle import pyoracle

STATIC = [
    (b":authority", b""), (b":method", b"GET"), (b":method", b"POST"), (b":path", b"/"), (b":path", b"/index.html"),
    (b":scheme", b"http"), (b":scheme", b"https"), (b":status", b"200"), (b":status", b"204"), (b":status", b"206"),
    (b":status", b"304"), (b":status", b"400"), (b":status", b"404"), (b":status", b"500"), (b"accept-charset", b""),
    (b"accept-encoding", b"gzip, deflate"), (b"accept-language", b""), (b"accept-ranges", b""), (b"accept", b""),
    (b"access-control-allow-origin", b""), (b"age", b""), (b"allow", b""), (b"authorization", b""), (b"cache-control", b""),
    (b"content-disposition", b""), (b"content-encoding", b""), (b"content-language", b""), (b"content-length", b""),
    (b"content-location", b""), (b"content-range", b""), (b"content-type", b""), (b"cookie", b""), (b"date", b""), (b"etag", b""),
    (b"expect", b""), (b"expires", b""), (b"from", b""), (b"host", b""), (b"if-match", b""), (b"if-modified-since", b""),
    (b"if-none-match", b""), (b"if-range", b""), (b"if-unmodified-since", b""), (b"last-modified", b""), (b"link", b""),
    (b"location", b""), (b"max-forwards", b""), (b"proxy-authenticate", b""), (b"proxy-authorization", b""), (b"range", b""),
    (b"referer", b""), (b"refresh", b""), (b"retry-after", b""), (b"server", b""), (b"set-cookie", b""),
    (b"strict-transport-security", b""), (b"transfer-encoding", b""), (b"user-agent", b""), (b"vary", b""), (b"via", b""),
    (b"www-authenticate", b"")]
assert len(STATIC) == 61


def varint(prefix_bits: int, value: int, first_byte_flags: int = 0) -> bytes:
    lim = (1 << prefix_bits) - 1
    if value < lim:
        return bytes([first_byte_flags | value])
    out = [first_byte_flags | lim]; value -= lim
    while value >= 128:
        out.append((value & 127) | 128); value >>= 7
    out.append(value)
    return bytes(out)


def string(s: bytes, huffman: bool) -> bytes:
    if huffman:
        h = pyoracle.huff_encode(s)
        return varint(7, len(h), 0x80) + h
    return varint(7, len(s), 0) + s


class Encoder:
    """A deliberately simple HPACK encoder that mirrors the decoder's table so that it can emit indexed fields."""

    def __init__(self, max_size: int = 4096):
        self.dyn: list = []; self.size = 0; self.max = max_size

    def _add(self, n: bytes, v: bytes):
        self.dyn.insert(0, (n, v)); self.size += len(n) + len(v) + 32
        while self.size > self.max and self.dyn:
            n2, v2 = self.dyn.pop(); self.size -= len(n2) + len(v2) + 32

    def _find(self, n: bytes, v: bytes) -> Tuple[int, int]:
        """-> (index of the exact field or 0, index of a field with that name or 0)"""
        full = name = 0
        for i, (sn, sv) in enumerate(STATIC + self.dyn, start=1):
            if sn == n:
                name = name or i
                if sv == v:
                    full = full or i
        return full, name

    def field(self, n: bytes, v: bytes, *, mode: str = "index", huffman: bool = False, use_index: bool = True) -> bytes:
        """mode: "index" (6.2.1) | "plain" (6.2.2) | "never" (6.2.3); use_index allows 6.1 / indexed names."""
        full, name = self._find(n, v) if use_index else (0, 0)
        if full:
            return varint(7, full, 0x80)
        prefix, flags = {"index": (6, 0x40), "plain": (4, 0x00), "never": (4, 0x10)}[mode]
        out = varint(prefix, name, flags) + (b"" if name else string(n, huffman)) + string(v, huffman)
        if mode == "index":
            self._add(n, v)
        return out

    def resize(self, new_max: int) -> bytes:
        self.max = new_max
        while self.size > self.max and self.dyn:
            n2, v2 = self.dyn.pop(); self.size -= len(n2) + len(v2) + 32
        return varint(5, new_max, 0x20)

    def block(self, fields: Iterable[Tuple[bytes, bytes]], **kw) -> bytes:
        return b"".join(self.field(n, v, **kw) for n, v in fields)


HEADERS, DATA, SETTINGS, WINDOW_UPDATE = 1, 0, 4, 8


def frame(ftype: int, stream: int, payload: bytes, flags: int = 0x4, length: Optional[int] = None) -> bytes:
    ln = len(payload) if length is None else length
    return ln.to_bytes(3, "big") + bytes([ftype, flags]) + stream.to_bytes(4, "big") + payload


def l7_record(pid: int, fd: int, method_id: int, payload: bytes, write_ns: int, saddr: int, daddr: int, *, tls: int = 0,
              sport: int = 40000, dport: int = 8080, proto: int = 4, status: int = 0, dur: int = 0) -> bytes:
    """One 1096-byte bpfL7Event (ebpf/l7_req/l7.go:345-369)."""
    r = bytearray(1096)
    r[0:8] = fd.to_bytes(8, "little"); r[8:16] = write_ns.to_bytes(8, "little"); r[16:20] = pid.to_bytes(4, "little")
    r[20:24] = status.to_bytes(4, "little"); r[24:32] = dur.to_bytes(8, "little"); r[32] = proto; r[33] = method_id
    payload = payload[:1024]
    r[36:36 + len(payload)] = payload; r[1060:1064] = len(payload).to_bytes(4, "little"); r[1064] = 1; r[1066] = tls
    r[1076:1080] = saddr.to_bytes(4, "little"); r[1080:1082] = sport.to_bytes(2, "little")
    r[1084:1088] = daddr.to_bytes(4, "little"); r[1088:1090] = dport.to_bytes(2, "little")
    return bytes(r)
