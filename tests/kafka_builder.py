"""Kafka wire builders for the tests (protocol guide, "Produce", "Fetch", "Record Batch", "Message sets").
Test infrastructure only.  Compression comes from pyarrow's codecs / zlib — implementations that are
independent of both decoders under test."""
from __future__ import annotations

import gzip as _gzip
import struct
import zlib
from typing import List, Optional, Sequence, Tuple

import pyarrow as pa

NONE, GZIP, SNAPPY, LZ4, ZSTD = 0, 1, 2, 3, 4
XERIAL_HEADER = bytes([130, 83, 78, 65, 80, 80, 89, 0]) + bytes([0, 0, 0, 1, 0, 0, 0, 1])


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    return c ^ 0xFFFFFFFF


def compress(codec: int, data: bytes, *, xerial: bool = False) -> bytes:
    if codec == NONE:
        return data
    if codec == GZIP:
        return _gzip.compress(data)
    if codec == SNAPPY:
        raw = pa.Codec("snappy").compress(data, asbytes=True)
        if not xerial:
            return raw
        half = len(data) // 2
        chunks = [pa.Codec("snappy").compress(c, asbytes=True) for c in (data[:half], data[half:])]
        return XERIAL_HEADER + b"".join(struct.pack(">I", len(c)) + c for c in chunks)
    if codec == LZ4:
        return pa.Codec("lz4").compress(data, asbytes=True)
    if codec == ZSTD:
        return pa.Codec("zstd").compress(data, asbytes=True)
    return data                                                       # unknown codec ids (5..7): stored as is, the decoder must refuse


def varint(v: int) -> bytes:
    u = (v << 1) ^ (v >> 63); u &= (1 << 64) - 1
    out = bytearray()
    while u >= 0x80:
        out.append((u & 0x7F) | 0x80); u >>= 7
    out.append(u)
    return bytes(out)


def vbytes(b: Optional[bytes]) -> bytes:
    return varint(-1) if b is None else varint(len(b)) + b


def string(s: Optional[bytes]) -> bytes:
    return struct.pack(">h", -1) if s is None else struct.pack(">h", len(s)) + s


def record(key: Optional[bytes], value: Optional[bytes], *, offset_delta: int = 0, ts_delta: int = 0,
           headers: Sequence[Tuple[bytes, bytes]] = (), length_override: Optional[bytes] = None) -> bytes:
    body = b"\x00" + varint(ts_delta) + varint(offset_delta) + vbytes(key) + vbytes(value) + varint(len(headers))
    for k, v in headers:
        body += vbytes(k) + vbytes(v)
    return (varint(len(body)) if length_override is None else length_override) + body


def record_batch(records: Sequence[bytes], *, codec: int = NONE, num_records: Optional[int] = None, base_offset: int = 0, magic: int = 2,
                 bad_crc: bool = False, xerial: bool = False, batch_len_delta: int = 0, attributes_extra: int = 0,
                 raw_payload: Optional[bytes] = None) -> bytes:
    payload = compress(codec, b"".join(records), xerial=xerial) if raw_payload is None else raw_payload
    n = len(records) if num_records is None else num_records
    after_crc = struct.pack(">hiqqqhii", codec | attributes_extra, max(n - 1, 0), 1_700_000_000_000, 1_700_000_000_000, -1, -1, -1, n) + payload
    crc = crc32c(after_crc) ^ (1 if bad_crc else 0)
    body = struct.pack(">ib", 0, magic) + struct.pack(">I", crc) + after_crc          # leader epoch, magic, crc, ...
    return struct.pack(">qi", base_offset, len(body) + batch_len_delta) + body


def legacy_message(key: Optional[bytes], value: Optional[bytes], *, magic: int = 1, codec: int = NONE, offset: int = 0) -> bytes:
    def b32(b):
        return struct.pack(">i", -1) if b is None else struct.pack(">i", len(b)) + b
    body = struct.pack(">bb", magic, codec) + (struct.pack(">q", 1_700_000_000_000) if magic == 1 else b"") + b32(key) + b32(value)
    msg = struct.pack(">I", zlib.crc32(body)) + body
    return struct.pack(">qi", offset, len(msg)) + msg


def produce_request(topics: Sequence[Tuple[Optional[bytes], Sequence[Tuple[int, bytes]]]], *, version: int = 7, correlation: int = 7,
                    client_id: Optional[bytes] = b"producer-1", transactional_id: Optional[bytes] = None, api_key: int = 0,
                    size_delta: int = 0, trailing: bytes = b"") -> bytes:
    """topics: [(name, [(partition, records-bytes), ...]), ...]"""
    body = struct.pack(">hhi", api_key, version, correlation) + string(client_id)
    if version >= 3:
        body += string(transactional_id)
    body += struct.pack(">hi", 1, 30000) + struct.pack(">i", len(topics))
    for name, parts in topics:
        body += string(name) + struct.pack(">i", len(parts))
        for pid, recs in parts:
            body += struct.pack(">ii", pid, len(recs)) + recs
    body += trailing
    return struct.pack(">i", len(body) + size_delta) + body


def fetch_response(topics: Sequence[Tuple[bytes, Sequence[Tuple[int, bytes]]]], *, version: int = 11, correlation: int = 7, size: Optional[int] = None,
                   aborted: int = 0) -> bytes:
    body = b""
    if version >= 1:
        body += struct.pack(">i", 0)                                  # throttle
    if version >= 7:
        body += struct.pack(">hi", 0, 0)                              # error code, session id
    body += struct.pack(">i", len(topics))
    for name, parts in topics:
        body += string(name) + struct.pack(">i", len(parts))
        for pid, recs in parts:
            body += struct.pack(">ihq", pid, 0, 100)                  # partition, error, high watermark
            if version >= 4:
                body += struct.pack(">q", 100)                        # last stable offset
                if version >= 5:
                    body += struct.pack(">q", 0)                      # log start offset
                body += struct.pack(">i", aborted) + b"".join(struct.pack(">qq", 1, 2) for _ in range(max(aborted, 0)))
            if version >= 11:
                body += struct.pack(">i", -1)                         # preferred read replica
            body += struct.pack(">i", len(recs)) + recs
    hdr = struct.pack(">i", correlation)
    return struct.pack(">i", (len(hdr) + len(body)) if size is None else size) + hdr + body


def l7_record(method_id: int, payload: bytes, write_ns: int, saddr: int, daddr: int, *, api_version: int = 0, pid: int = 77, fd: int = 9,
              dur: int = 1000, tls: int = 0) -> bytes:
    r = bytearray(1096)
    r[0:8] = fd.to_bytes(8, "little"); r[8:16] = write_ns.to_bytes(8, "little"); r[16:20] = pid.to_bytes(4, "little")
    r[20:24] = (1).to_bytes(4, "little"); r[24:32] = dur.to_bytes(8, "little"); r[32] = 6; r[33] = method_id
    payload = payload[:1024]
    r[36:36 + len(payload)] = payload; r[1060:1064] = len(payload).to_bytes(4, "little"); r[1064] = 1; r[1066] = tls
    r[1068:1070] = api_version.to_bytes(2, "little", signed=True)
    r[1076:1080] = saddr.to_bytes(4, "little"); r[1080:1082] = (40000).to_bytes(2, "little")
    r[1084:1088] = daddr.to_bytes(4, "little"); r[1088:1090] = (9092).to_bytes(2, "little")
    return bytes(r)
