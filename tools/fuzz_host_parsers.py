#!/usr/bin/env python
"""Mutation fuzzing of the host-side parsers, C++ product vs C oracle, meant to run under the sanitizer builds
(tools/asan_host_tests.sh runs it last).  Every input goes through both implementations; results must agree and neither
may trip ASan/UBSan.  usage: fuzz_host_parsers.py [iterations] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alaz_amd import hostlib
from oracle import pyoracle
from tests import h2_builder as hb, kafka_builder as kb

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def mutate(p: bytes) -> bytes:
    b = bytearray(p)
    for _ in range(rng.randrange(1, 4)):
        r = rng.random()
        if not b:
            b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        elif r < 0.35:
            i = rng.randrange(len(b)); b[i] ^= 1 << rng.randrange(8)
        elif r < 0.5:
            i = rng.randrange(len(b)); b[i] = rng.choice([0, 0xFF, 0x7F, 0x80, rng.randrange(256)])
        elif r < 0.65:
            del b[rng.randrange(len(b)):]
        elif r < 0.75:
            i = rng.randrange(len(b)); del b[i:i + rng.randrange(1, 6)]
        elif r < 0.85:
            i = rng.randrange(len(b) + 1); b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 6)))
        elif r < 0.95 and len(b) >= 4:                                   # overwrite a big-endian length-looking field
            i = rng.randrange(len(b) - 3); b[i:i + 4] = rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, rng.randrange(1 << 16)]).to_bytes(4, "big")
        else:
            i = rng.randrange(len(b)); j = rng.randrange(len(b)); b[i:i] = b[j:j + rng.randrange(1, 16)]
    return bytes(b[:1024])


def kafka_seed():
    recs = [kb.record(bytes(rng.randrange(256) for _ in range(rng.randrange(0, 5))) if rng.random() < 0.8 else None,
                      bytes(rng.randrange(256) for _ in range(rng.randrange(0, 24))) if rng.random() < 0.9 else None, offset_delta=i,
                      headers=[(b"h", b"v")] * rng.randrange(0, 3)) for i in range(rng.randrange(0, 5))]
    codec = rng.choice([0, 0, 1, 2, 3, 4])
    body = kb.record_batch(recs, codec=codec, xerial=rng.random() < 0.5) if rng.random() < 0.9 else kb.legacy_message(b"k", kb.compress(codec, kb.legacy_message(b"a", b"b")) if codec else b"v", codec=codec)
    topics = [(rng.choice([b"a", b"topic"]), [(rng.randrange(3), body)] * rng.randrange(1, 3))] * rng.randrange(1, 3)
    if rng.random() < 0.5:
        return 1, rng.choice([0, 2, 3, 7]), kb.produce_request(topics, version=rng.choice([2, 3, 7]))
    v = rng.randrange(0, 13)
    return 2, v, kb.fetch_response(topics, version=min(v, 11), aborted=rng.randrange(0, 2) if v >= 4 else 0)


def main():
    stats = {"kafka": 0, "kafka_ok": 0, "codec": 0, "hpack": 0, "huffman": 0, "h2": 0, "wire": 0, "procline": 0}
    # (the reference prints "%d" of a signed parse: a pair like "-8" gives the text "-8"; the numeric product keeps its low byte)
    ipn = lambda s_: sum((int(x) & 255) << (24 - 8 * i) for i, x in enumerate(s_.split(".")))
    for it in range(N):
        # Kafka payloads
        m, v, p = kafka_seed()
        p = mutate(p) if rng.random() < 0.9 else p
        a, b = pyoracle.kafka_decode(p, m, v), hostlib.kafka_decode(p, m, v)
        assert a == b, ("kafka", it, p.hex())
        stats["kafka"] += 1; stats["kafka_ok"] += a[0] == "ok" and bool(a[1])
        # raw decompressors
        codec = rng.randrange(1, 5); data = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 200))) * rng.randrange(1, 4)
        c = mutate(kb.compress(codec, data, xerial=rng.random() < 0.5))
        assert pyoracle.kafka_decompress(codec, c) == hostlib.kafka_decompress(codec, c), ("codec", codec, c.hex()); stats["codec"] += 1
        # HPACK blocks through fresh decoders, in two pieces
        enc = hb.Encoder()
        blk = b"".join(enc.field(rng.choice([b":method", b":path", b"x-a", b"content-type"]), bytes(rng.randrange(256) for _ in range(rng.randrange(0, 12))),
                                 mode=rng.choice(["index", "plain", "never"]), huffman=rng.random() < 0.5) for _ in range(rng.randrange(1, 6)))
        blk = mutate(blk); cut = rng.randrange(len(blk) + 1)
        da, dbb = pyoracle.Hpack(), hostlib.Hpack()
        for piece in (blk[:cut], blk[cut:]):
            assert da.write(piece) == dbb.write(piece), ("hpack", blk.hex(), cut)
        assert da.table() == dbb.table(); stats["hpack"] += 1
        h = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 10)))
        assert pyoracle.huff_decode(h) == hostlib.huffman_decode(h); stats["huffman"] += 1
        # HTTP/2 events through both assemblers (same mutated frame to both)
        if it % 50 == 0:
            A, B = pyoracle.H2Assembler(), hostlib.Http2Assembler(); A.proc_exec(1); B.proc_exec(1); e1, e2 = hb.Encoder(), hb.Encoder(); sid = 1
        side = rng.choice([1, 2])
        fr = hb.frame(hb.HEADERS, sid, (e1 if side == 1 else e2).block([(b":method", b"GET"), (b":path", b"/p%d" % (it % 5)), (b":authority", b"a")] if side == 1 else [(b":status", b"200")]),
                      flags=rng.choice([4, 5, 0x24, 0x2C]))
        fr = mutate(fr) if rng.random() < 0.6 else fr
        ra, rb = A.event(1, 2, side, fr, 1000 + it, False), B.event(1, 2, side, fr, 1000 + it, False)
        assert ra == rb and A.pending() == B.pending(), ("h2", it, fr.hex()); stats["h2"] += 1; sid += 2 * (side == 2)
        # whole wire records through the packer (HTTP/1, SQL, Mongo, Redis ... payload handlers) — product only + oracle row count
        if it % 10 == 0:
            rec = bytearray(hb.l7_record(5, 6, rng.randrange(0, 5), mutate(rng.choice([b"GET /a HTTP/1.1\r\nHost: x.y\r\n\r\n", b"Q\x00\x00\x00\x0dselect 1\x00",
                                         b"P\x00\x00\x00\x10s1\x00select 2\x00\x00\x00", b"\x03select 3", b"*1\r\n$4\r\nPING\r\n", b"\x10\x00\x00\x00" * 8])),
                                         100 + it, 0x0A000001, 0x08080808, proto=rng.randrange(0, 10)))
            rec[1060:1064] = rng.choice([len(rec), 0, 3, 5, 1024, 4096, 0xFFFFFFFF]).to_bytes(4, "little") if rng.random() < 0.3 else rec[1060:1064]
            pk = hostlib.Packer(); pk.proc_exec(5); pk.kafka_decode(True); pk.pack_wire(bytes(rec)); pk.pack_wire(bytes(rec), full_copy=True)
            o = pyoracle.Oracle(0, 0); o.pod("ADD", "p", "10.0.0.1"); o.set_kafka_decode(True); o.h2().proc_exec(5); o.l7_wire(bytes(rec))
            stats["wire"] += 1
        # /proc/<pid>/net/tcp lines and fd link texts (the socket-line seeding, sock_num_line.go:351-397): ASCII-ish mutations
        line = mutate(b"   3: 7038A8C0:A24A C28D640A:0050 01 00000000:00000000 02:000002E0 00000000  1000        0 5276530 2 ffff8e8be7a0bd40 20 4 24 10 -1")
        line = bytes(c for c in line if c not in (0,) and c < 128).decode("ascii")
        a, b = pyoracle.parse_tcp_line(line), hostlib.proc_parse_tcp_line(line)
        assert (a is None) == (b is None) and (a is None or (ipn(a[0]), a[1], ipn(a[2]), a[3]) == b), ("procline", line)
        link = bytes(c for c in mutate(b"socket:[5276530]") if c != 0 and c < 128).decode("ascii")
        assert pyoracle.inode_from_link(link) == hostlib.proc_inode_of_link(link), ("link", link)
        stats["procline"] += 1
    print("fuzz ok:", stats)


if __name__ == "__main__":
    main()
