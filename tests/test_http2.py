"""SURVEY.md §8 f-4 (first half): HTTP/2 request assembly on the host.

* The oracle's HPACK restatement (oracle/http2.c) and the C++ decoder of the product's host side
  (alaz_amd/csrc/host/http2.cpp) against the known-answer vectors of RFC 7541 Appendix C — the reference
  delegates HPACK to golang.org/x/net v0.20.0 (not vendored) and holds no test of its own for this path.
* C++ vs oracle differentially: Huffman strings, random header blocks split at arbitrary points and
  corrupted, strconv.Atoi corner cases, random frame traces through both assemblers.
* The reference's frame-assembly rules (aggregator/data.go:544-810) as scenario tests.
* HTTP/2 records through the packer -> packed events -> same edges as the oracle's wire path.
All CPU-only."""
import random

import numpy as np
import pytest

from alaz_amd import build, engine, hostlib, replay, weights
from oracle import pyoracle
from tests import h2_builder as hb

H = lambda s: bytes.fromhex(s.replace(" ", "").replace("\n", ""))
CLOCK = (1_000_000_000, 1_700_000_000_000_000_000)


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build_all()


DECODERS = [("oracle", pyoracle.Hpack), ("host", hostlib.Hpack)]

# ------------------------------------------------------------------------------------------------ RFC 7541 Appendix C
RFC_STRINGS = [
    (b"www.example.com", "f1e3 c2e5 f23a 6ba0 ab90 f4ff"), (b"no-cache", "a8eb 1064 9cbf"),
    (b"custom-key", "25a8 49e9 5ba9 7d7f"), (b"custom-value", "25a8 49e9 5bb8 e8b4 bf"),
    (b"302", "6402"), (b"private", "aec3 771a 4b"), (b"307", "640e ff"), (b"gzip", "9bd9 ab"),
    (b"Mon, 21 Oct 2013 20:13:21 GMT", "d07a be94 1054 d444 a820 0595 040b 8166 e082 a62d 1bff"),
    (b"Mon, 21 Oct 2013 20:13:22 GMT", "d07a be94 1054 d444 a820 0595 040b 8166 e084 a62d 1bff"),
    (b"https://www.example.com", "9d29 ad17 1863 c78f 0b97 c8e9 ae82 ae43 d3"),
    (b"foo=ASDJKHQKBZXOQWEOPIUAXQWEOIU; max-age=3600; version=1",
     "94e7 821d d7f2 e6c7 b335 dfdf cd5b 3960 d5af 2708 7f36 72c1 ab27 0fb5 291f 9587 3160 65c0 03ed 4ee5 b106 3d50 07"),
]


def test_huffman_table_is_a_complete_prefix_code():
    assert pyoracle.lib().or_hpack_selfcheck() == 0
    # a few published codes (RFC 7541 Appendix B)
    for sym, code, ln in [(ord("0"), 0x0, 5), (ord("t"), 0x9, 5), (ord(" "), 0x14, 6), (ord("u"), 0x2D, 6), (ord(":"), 0x5C, 7),
                          (ord("z"), 0x7B, 7), (ord("&"), 0xF8, 8), (ord("Z"), 0xFD, 8), (ord("!"), 0x3F8, 10), (ord("?"), 0x3FC, 10),
                          (ord("'"), 0x7FA, 11), (ord("#"), 0xFFA, 12), (0, 0x1FF8, 13), (ord("^"), 0x3FFC, 14), (ord("<"), 0x7FFC, 15),
                          (ord("\\"), 0x7FFF0, 19), (128, 0xFFFE6, 20), (256, 0x3FFFFFFF, 30)]:
        assert pyoracle.huff_code(sym) == (code, ln), sym


def test_huffman_table_equals_the_rfc_appendix_fixture():
    """All 257 codes of RFC 7541 Appendix B (tests/golden/hpack_huffman_rfc7541.json), on both implementations."""
    import json, os
    tab = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hpack_huffman_rfc7541.json")))["table"]
    assert len(tab) == 257
    for sym, (code, ln) in enumerate(tab):
        assert pyoracle.huff_code(sym) == (code, ln)
        if sym < 256:                                  # the C++ side through its encoder: the code, padded with ones
            pad = (-ln) % 8
            want = ((code << pad) | ((1 << pad) - 1)).to_bytes((ln + pad) // 8, "big")
            assert hostlib.huffman_encode(bytes([sym])) == want
            if pad < 8:
                assert hostlib.huffman_decode(want) == bytes([sym]) == pyoracle.huff_decode(want)


@pytest.mark.parametrize("plain,hexed", RFC_STRINGS)
def test_huffman_known_answers(plain, hexed):
    enc = H(hexed)
    assert pyoracle.huff_encode(plain) == enc and hostlib.huffman_encode(plain) == enc
    assert pyoracle.huff_decode(enc) == plain and hostlib.huffman_decode(enc) == plain


DATE1, DATE2 = b"Mon, 21 Oct 2013 20:13:21 GMT", b"Mon, 21 Oct 2013 20:13:22 GMT"
COOKIE = b"foo=ASDJKHQKBZXOQWEOPIUAXQWEOIU; max-age=3600; version=1"
LOC = b"https://www.example.com"
REQ1 = [(b":method", b"GET"), (b":scheme", b"http"), (b":path", b"/"), (b":authority", b"www.example.com")]
REQ2 = REQ1 + [(b"cache-control", b"no-cache")]
REQ3 = [(b":method", b"GET"), (b":scheme", b"https"), (b":path", b"/index.html"), (b":authority", b"www.example.com"), (b"custom-key", b"custom-value")]
RSP1 = [(b":status", b"302"), (b"cache-control", b"private"), (b"date", DATE1), (b"location", LOC)]
RSP2 = [(b":status", b"307"), (b"cache-control", b"private"), (b"date", DATE1), (b"location", LOC)]
RSP3 = [(b":status", b"200"), (b"cache-control", b"private"), (b"date", DATE2), (b"location", LOC), (b"content-encoding", b"gzip"), (b"set-cookie", COOKIE)]
T_REQ = [([(b":authority", b"www.example.com")], 57),
         ([(b"cache-control", b"no-cache"), (b":authority", b"www.example.com")], 110),
         ([(b"custom-key", b"custom-value"), (b"cache-control", b"no-cache"), (b":authority", b"www.example.com")], 164)]
T_RSP = [([(b"location", LOC), (b"date", DATE1), (b"cache-control", b"private"), (b":status", b"302")], 222),
         ([(b":status", b"307"), (b"location", LOC), (b"date", DATE1), (b"cache-control", b"private")], 222),
         ([(b"set-cookie", COOKIE), (b"content-encoding", b"gzip"), (b"date", DATE2)], 215)]

SEQUENCES = {
    "C.3 requests, no huffman": (4096, [
        "8286 8441 0f77 7777 2e65 7861 6d70 6c65 2e63 6f6d",
        "8286 84be 5808 6e6f 2d63 6163 6865",
        "8287 85bf 400a 6375 7374 6f6d 2d6b 6579 0c63 7573 746f 6d2d 7661 6c75 65"], [REQ1, REQ2, REQ3], T_REQ),
    "C.4 requests, huffman": (4096, [
        "8286 8441 8cf1 e3c2 e5f2 3a6b a0ab 90f4 ff",
        "8286 84be 5886 a8eb 1064 9cbf",
        "8287 85bf 4088 25a8 49e9 5ba9 7d7f 8925 a849 e95b b8e8 b4bf"], [REQ1, REQ2, REQ3], T_REQ),
    "C.5 responses, no huffman, 256-byte table": (256, [
        "4803 3330 3258 0770 7269 7661 7465 611d 4d6f 6e2c 2032 3120 4f63 7420 3230 3133 2032 303a 3133 3a32 3120 474d 546e 1768"
        "7474 7073 3a2f 2f77 7777 2e65 7861 6d70 6c65 2e63 6f6d",
        "4803 3330 37c1 c0bf",
        "88c1 611d 4d6f 6e2c 2032 3120 4f63 7420 3230 3133 2032 303a 3133 3a32 3220 474d 54c0 5a04 677a 6970 7738 666f 6f3d 4153"
        "444a 4b48 514b 425a 584f 5157 454f 5049 5541 5851 5745 4f49 553b 206d 6178 2d61 6765 3d33 3630 303b 2076 6572 7369 6f6e 3d31"],
        [RSP1, RSP2, RSP3], T_RSP),
    "C.6 responses, huffman, 256-byte table": (256, [
        "4882 6402 5885 aec3 771a 4b61 96d0 7abe 9410 54d4 44a8 2005 9504 0b81 66e0 82a6 2d1b ff6e 919d 29ad 1718 63c7 8f0b 97c8"
        "e9ae 82ae 43d3",
        "4883 640e ffc1 c0bf",
        "88c1 6196 d07a be94 1054 d444 a820 0595 040b 8166 e084 a62d 1bff c05a 839b d9ab 77ad 94e7 821d d7f2 e6c7 b335 dfdf cd5b"
        "3960 d5af 2708 7f36 72c1 ab27 0fb5 291f 9587 3160 65c0 03ed 4ee5 b106 3d50 07"],
        [RSP1, RSP2, RSP3], T_RSP),
}


@pytest.mark.parametrize("who,cls", DECODERS)
@pytest.mark.parametrize("name", list(SEQUENCES))
def test_rfc7541_appendix_c_sequences(who, cls, name):
    """Header lists AND dynamic-table contents/sizes after every block, exactly as the RFC prints them."""
    max_size, blocks, want_fields, want_tables = SEQUENCES[name]
    d = cls(max_size)
    for blk, fields, (table, size) in zip(blocks, want_fields, want_tables):
        rc, got = d.write(H(blk))
        assert rc == 0 and got == fields
        assert d.table() == table and d.table_size() == size


@pytest.mark.parametrize("who,cls", DECODERS)
def test_static_table_is_rfc7541_appendix_a(who, cls):
    """Indexed fields 1..61 against the builder's copy of Appendix A (itself checked once against python-hpack's table)."""
    d = cls()
    assert [d.write(bytes([0x80 | i]))[1][0] for i in range(1, 62)] == hb.STATIC
    assert d.write(bytes([0x80 | 62]))[0] == -1 and d.table() == []


@pytest.mark.parametrize("who,cls", DECODERS)
def test_rfc7541_c2_single_field_representations(who, cls):
    d = cls()
    assert d.write(H("400a 6375 7374 6f6d 2d6b 6579 0d63 7573 746f 6d2d 6865 6164 6572")) == (0, [(b"custom-key", b"custom-header")])
    assert d.table() == [(b"custom-key", b"custom-header")] and d.table_size() == 55
    d = cls()
    assert d.write(H("040c 2f73 616d 706c 652f 7061 7468")) == (0, [(b":path", b"/sample/path")]) and d.table() == []
    assert d.write(H("1008 7061 7373 776f 7264 0673 6563 7265 74")) == (0, [(b"password", b"secret")]) and d.table() == []
    assert d.write(H("82")) == (0, [(b":method", b"GET")])


@pytest.mark.parametrize("who,cls", DECODERS)
def test_hpack_write_semantics_of_the_go_decoder(who, cls):
    """golang.org/x/net hpack.Decoder.Write as the reference uses it (never Close()d between blocks)."""
    # a block cut inside a field is kept and completed by the next Write — whatever that Write belongs to
    d = cls(); blk = H("8286 8441 0f77 7777 2e65 7861 6d70 6c65 2e63 6f6d")
    assert d.write(blk[:7]) == (0, REQ1[:3])
    assert d.write(blk[7:]) == (0, REQ1[3:]) and d.table_size() == 57
    # index 0 and an index beyond the table are decoding errors; fields before the error were emitted; the rest
    # of that Write is dropped and the decoder stays usable
    d = cls()
    assert d.write(H("82 80 84")) == (-1, [(b":method", b"GET")])
    assert d.write(H("82 be 84")) == (-1, [(b":method", b"GET")])
    assert d.write(H("84")) == (0, [(b":path", b"/")])
    # a size update is accepted only as the connection's very first field or while the table is empty
    d = cls(); assert d.write(H("3f e1 1f")) == (0, []) and d.write(H("20 82")) == (0, [(b":method", b"GET")])     # 4096 first, then 0: table empty
    d = cls(); d.write(H("400a 6375 7374 6f6d 2d6b 6579 0d63 7573 746f 6d2d 6865 6164 6572"))
    assert d.write(H("20 82")) == (-1, [])                                                                         # table non-empty, not first
    d = cls(); assert d.write(H("3f e2 1f")) == (-1, [])                                                           # 4097 > allowed maximum
    # invalid huffman: EOS inside, padding with a zero bit, 8 bits of padding
    for bad in ("00 84 ffff ffff 00", "00 81 1e 00", "00 82 1f ff 00"):
        d = cls(); assert d.write(H(bad))[0] == -1, bad
    # varint overflow
    d = cls(); assert d.write(H("ff ff ff ff ff ff ff ff ff ff ff 01"))[0] == -1
    # unknown representation cannot occur (all 256 first bytes are covered): every first byte parses or asks for more
    for b in range(256):
        d = cls(); rc, _ = d.write(bytes([b])); assert rc in (0, -1)


def test_hpack_table_shrink_details():
    for _, cls in DECODERS:
        d = cls()
        assert d.write(H("3f 19") + H("400a 6375 7374 6f6d 2d6b 6579 0d63 7573 746f 6d2d 6865 6164 6572"))[0] == 0   # max = 31+25 = 56 >= 55
        assert d.table_size() == 55
        d = cls()
        assert d.write(H("3f 17") + H("400a 6375 7374 6f6d 2d6b 6579 0d63 7573 746f 6d2d 6865 6164 6572"))[0] == 0   # max = 54 < 55
        assert d.table_size() == 0 and d.table() == []


@pytest.mark.parametrize("v", [b"", b"0", b"200", b"+7", b"-7", b"+", b"-", b"12a", b" 12", b"0012", b"4294967296", b"4294967295",
                               b"9223372036854775807", b"9223372036854775808", b"-9223372036854775808", b"-9223372036854775809",
                               b"99999999999999999999999", b"1_000", b"0x10", b"\xff"])
def test_go_atoi_semantics(v):
    """uint32(s) after `s, _ := strconv.Atoi(v)` (data.go:783-789): syntax error -> 0, range error -> clamped int64."""
    def model(b):
        try:
            t = b.decode("ascii")
        except UnicodeDecodeError:
            return 0
        body = t[1:] if t[:1] in "+-" else t
        if not body or not all(c in "0123456789" for c in body):
            return 0
        n = int(t); n = max(-(1 << 63), min((1 << 63) - 1, n))
        return n & 0xFFFFFFFF
    assert pyoracle.go_atoi_u32(v) == hostlib.go_atoi_u32(v) == model(v)


# ------------------------------------------------------------------------------------------------ differential fuzz
def test_huffman_differential_random_strings():
    rng = random.Random(7)
    for _ in range(400):
        s = bytes(rng.randrange(256) if rng.random() < 0.2 else rng.choice(b"abcdefghijklmnopqrstuvwxyz0123456789-./:=_% ") for _ in range(rng.randrange(0, 60)))
        e = pyoracle.huff_encode(s)
        assert hostlib.huffman_encode(s) == e
        assert pyoracle.huff_decode(e) == s and hostlib.huffman_decode(e) == s
    for _ in range(3000):                      # arbitrary bytes: both accept or both reject, same text
        b = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 12)))
        assert pyoracle.huff_decode(b) == hostlib.huffman_decode(b), b.hex()
    for n in range(0, 6):                      # every tail of ones
        b = b"\xff" * n
        assert pyoracle.huff_decode(b) == hostlib.huffman_decode(b)


NAMES = [b":method", b":path", b":authority", b"content-type", b":status", b"grpc-status", b"x-request-id", b"user-agent", b"te", b"x-b3-traceid"]
VALUES = [b"GET", b"POST", b"/", b"/api/v1/items", b"/grpc.Service/Method", b"svc.ns.svc.cluster.local:8080", b"application/grpc",
          b"application/json", b"200", b"503", b"0", b"14", b"trailers", b"curl/8.0", b"a" * 70, b"\x00\xff binary \x80"]


def _random_blocks(rng, n_blocks):
    enc = hb.Encoder()
    for _ in range(n_blocks):
        parts = []
        for _ in range(rng.randrange(1, 7)):
            r = rng.random()
            if r < 0.04:
                parts.append(enc.resize(rng.choice([0, 64, 256, 4096])))      # mostly an error for the Go decoder
            else:
                n = rng.choice(NAMES); v = rng.choice(VALUES) if rng.random() < 0.8 else bytes(rng.randrange(32, 127) for _ in range(rng.randrange(0, 20)))
                parts.append(enc.field(n, v, mode=rng.choice(["index", "index", "plain", "never"]), huffman=rng.random() < 0.5, use_index=rng.random() < 0.8))
        yield b"".join(parts)


def test_hpack_differential_random_blocks_splits_and_corruption():
    rng = random.Random(2024)
    for trial in range(150):
        a, b = pyoracle.Hpack(), hostlib.Hpack()
        for blk in _random_blocks(rng, rng.randrange(1, 8)):
            if rng.random() < 0.15 and blk:
                i = rng.randrange(len(blk)); blk = blk[:i] + bytes([blk[i] ^ (1 << rng.randrange(8))]) + blk[i + 1:]
            cuts = sorted(rng.sample(range(len(blk) + 1), min(len(blk) + 1, rng.randrange(0, 3))))
            pieces = [blk[i:j] for i, j in zip([0] + cuts, cuts + [len(blk)])]
            for p in pieces:
                ra, rb = a.write(p), b.write(p)
                assert ra == rb, (trial, p.hex())
            assert a.table() == b.table() and a.table_size() == b.table_size()


# ------------------------------------------------------------------------------------------------ assembler scenarios
ASSEMBLERS = [("oracle", pyoracle.H2Assembler), ("host", hostlib.Http2Assembler)]
CLIENT, SERVER = 1, 2


def _req(enc, method=b"POST", path=b"/svc/Do", authority=b"backend:8080", ctype=None, **kw):
    f = [(b":method", method), (b":scheme", b"http"), (b":path", path), (b":authority", authority)]
    if ctype:
        f.append((b"content-type", ctype))
    return enc.block(f, **kw)


@pytest.mark.parametrize("who,cls", ASSEMBLERS)
def test_assembler_pairs_the_two_headers_frames_of_a_stream(who, cls):
    a = cls(); a.proc_exec(10)
    ce, se = hb.Encoder(), hb.Encoder()
    # SETTINGS + WINDOW_UPDATE before HEADERS: skipped (data.go:683-686)
    pl = hb.frame(hb.SETTINGS, 0, b"\x00\x03\x00\x00\x00\x64", flags=0) + hb.frame(hb.WINDOW_UPDATE, 0, b"\x00\x0f\x00\x01", flags=0) + hb.frame(hb.HEADERS, 1, _req(ce))
    assert a.event(10, 5, CLIENT, pl, 1000) is None and a.pending() == 1 and a.parsers() == 1
    got = a.event(10, 5, SERVER, hb.frame(hb.HEADERS, 1, se.block([(b":status", b"503")])), 1750)
    assert got == (b"POST", b"/svc/Do", b"backend:8080", b"HTTP2", 503, 750) and a.pending() == 0
    # server first, client second: latency is 0 (req.Latency is overwritten with the client's own write time, :702)
    assert a.event(10, 5, SERVER, hb.frame(hb.HEADERS, 3, se.block([(b":status", b"200")])), 5000) is None
    assert a.event(10, 5, CLIENT, hb.frame(hb.HEADERS, 3, _req(ce)), 4000, True) == (b"POST", b"/svc/Do", b"backend:8080", b"HTTPS", 200, 0)
    # gRPC: content-type prefix, status = grpc-status of the first server HEADERS frame that carries it
    assert a.event(10, 5, CLIENT, hb.frame(hb.HEADERS, 5, _req(ce, ctype=b"application/grpc+proto")), 9000) is None
    got = a.event(10, 5, SERVER, hb.frame(hb.HEADERS, 5, se.block([(b":status", b"200"), (b"grpc-status", b"14")])), 9900, True)
    assert got == (b"POST", b"/svc/Do", b"backend:8080", b"gRPC", 14, 900)
    # streams are keyed by (pid, fd, stream): same stream id on another fd does not pair
    assert a.event(10, 6, SERVER, hb.frame(hb.HEADERS, 7, hb.Encoder().block([(b":status", b"200")])), 100) is None
    assert a.event(10, 5, CLIENT, hb.frame(hb.HEADERS, 7, _req(ce)), 100) is None and a.pending() == 2 and a.parsers() == 2


@pytest.mark.parametrize("who,cls", ASSEMBLERS)
def test_assembler_reference_quirks(who, cls):
    a = cls()
    ce, se = hb.Encoder(), hb.Encoder()
    # events of pids that are not live are dropped before any state exists (:1023-1029)
    assert a.event(10, 5, CLIENT, hb.frame(hb.HEADERS, 1, _req(hb.Encoder())), 1) is None and a.parsers() == 0 and a.pending() == 0
    a.proc_exec(10); a.proc_exec(1); a.proc_exec(100)
    # only the FIRST HEADERS frame of an event is looked at (:741, :800): stream 3 is never seen
    two = hb.frame(hb.HEADERS, 1, _req(ce)) + hb.frame(hb.HEADERS, 3, _req(ce))
    assert a.event(10, 5, CLIENT, two, 10) is None and a.pending() == 1
    # a frame cut by the 1 KiB capture stops the walk (:676-678); fewer than 9 bytes left too
    assert a.event(10, 5, SERVER, hb.frame(hb.HEADERS, 1, se.block([(b":status", b"200")]), length=500), 20) is None
    assert a.event(10, 5, SERVER, b"\x00\x00\x04\x01\x04\x00\x00", 20) is None and a.pending() == 1
    # the server write time precedes the client's: uint64 difference wraps, request ignored (:608-611)
    assert a.event(10, 5, SERVER, hb.frame(hb.HEADERS, 1, hb.Encoder().block([(b":status", b"200")])), 5) is None and a.pending() == 0
    # a request whose :method/:path could not be decoded is discarded (:577-585): a NEW parser (after conn close)
    # cannot resolve the dynamic-table references of the old connection's encoder
    assert a.event(10, 5, CLIENT, hb.frame(hb.HEADERS, 9, _req(ce, path=b"/long/unique/path")), 100) is None     # literal + indexed
    assert a.event(10, 5, SERVER, hb.frame(hb.HEADERS, 9, se.block([(b":status", b"200")])), 200) == (b"POST", b"/long/unique/path", b"backend:8080", b"HTTP2", 200, 100)
    a.conn_closed(10, 5); assert a.parsers() == 0
    blk = _req(ce, path=b"/long/unique/path")                      # now fully indexed into the (lost) dynamic table
    assert a.event(10, 5, CLIENT, hb.frame(hb.HEADERS, 11, blk), 300) is None
    assert a.event(10, 5, SERVER, hb.frame(hb.HEADERS, 11, hb.Encoder().block([(b":status", b"200")])), 400) is None and a.pending() == 0
    # the minute sweep forgets streams with one side only (:553-567)
    assert a.event(10, 7, CLIENT, hb.frame(hb.HEADERS, 1, _req(hb.Encoder())), 1) is None and a.pending() == 1
    a.sweep(); assert a.pending() == 0
    # processExit drops parsers by string prefix of the pid: exit of pid 1 also drops pid 10's and pid 100's (:362-377)
    for pid in (1, 10, 100, 2):
        a.proc_exec(pid); a.event(pid, 3, CLIENT, hb.frame(hb.DATA, 1, b"xx"), 1)
    assert a.parsers() == 6; a.proc_exit(1); assert a.parsers() == 1          # (10,5) (10,7) (1,3) (10,3) (100,3) go; (2,3) stays
    # PADDED|PRIORITY flags are not honoured: the 5 priority bytes go to the HPACK decoder as they are (:731)
    a.proc_exec(20)
    prio = b"\x00\x00\x00\x00\x10"       # 0x00 x4: four "literal, new name" starts... garbage by design
    r1 = a.event(20, 1, CLIENT, hb.frame(hb.HEADERS, 1, prio + _req(hb.Encoder()), flags=0x24), 1)
    r2 = a.event(20, 1, SERVER, hb.frame(hb.HEADERS, 1, hb.Encoder().block([(b":status", b"200")])), 2)
    assert r1 is None and r2 is None      # the request block was swallowed as literals: no :method / :path


def test_assembler_differential_random_traces():
    rng = random.Random(99)
    for trial in range(40):
        a, b = pyoracle.H2Assembler(), hostlib.Http2Assembler()
        pids = [7, 70, 71, 8]
        for p in pids[:3]:
            a.proc_exec(p); b.proc_exec(p)
        encs = {}
        t = 1000
        for step in range(200):
            pid = rng.choice(pids); fd = rng.randrange(3); side = rng.choice([CLIENT, SERVER, 3]); stream = rng.randrange(1, 8) * 2 - 1
            enc = encs.setdefault((pid, fd, side), hb.Encoder())
            frames = []
            for _ in range(rng.randrange(0, 3)):
                frames.append(hb.frame(rng.choice([hb.DATA, hb.SETTINGS, hb.WINDOW_UPDATE]), stream, bytes(rng.randrange(256) for _ in range(rng.randrange(0, 12))), flags=0))
            if rng.random() < 0.85:
                if side == SERVER:
                    f = [(b":status", rng.choice([b"200", b"404", b"500", b"x"]))] + ([(b"grpc-status", rng.choice([b"0", b"13", b""]))] if rng.random() < 0.4 else [])
                else:
                    f = [(b":method", rng.choice([b"GET", b"POST", b""])), (b":path", rng.choice([b"/", b"/a/b", b"/x.Y/Z"])), (b":authority", rng.choice([b"svc-a", b"svc-b:80", b""]))]
                    if rng.random() < 0.4:
                        f.append((b"content-type", rng.choice([b"application/grpc", b"application/grpc+json", b"application/json", b"application/grp"])))
                blk = enc.block(f, mode=rng.choice(["index", "plain"]), huffman=rng.random() < 0.5)
                if rng.random() < 0.1 and len(blk) > 2:
                    blk = blk[:rng.randrange(1, len(blk))]                     # cut block: the tail is carried to the next Write
                frames.append(hb.frame(hb.HEADERS, stream, blk, flags=rng.choice([0x4, 0x5, 0x24])))
            payload = b"".join(frames)
            if rng.random() < 0.05:
                payload = payload[:rng.randrange(0, len(payload) + 1)]
            t += rng.randrange(-50, 400)
            tls = rng.random() < 0.3
            ra, rb = a.event(pid, fd, side, payload, t, tls), b.event(pid, fd, side, payload, t, tls)
            assert ra == rb, (trial, step)
            r = rng.random()
            if r < 0.02:
                a.sweep(); b.sweep()
            elif r < 0.04:
                a.conn_closed(pid, fd); b.conn_closed(pid, fd); encs.pop((pid, fd, CLIENT), None); encs.pop((pid, fd, SERVER), None)
            elif r < 0.05:
                a.proc_exit(7); b.proc_exit(7); a.proc_exec(7); b.proc_exec(7)
            assert a.pending() == b.pending() and a.parsers() == b.parsers()


# ------------------------------------------------------------------------------------------------ through the packer
def _h2_trace(topo, n_streams, seed):
    """wire records of n_streams request/response pairs over a few connections between known pods, services and
    outbound hosts (named by :authority), interleaved with plain HTTP events."""
    rng = random.Random(seed)
    pods = [int(x) for x in topo.pod_ips[:20]]; svcs = [int(x) for x in topo.svc_ips[:8]]
    ext = [0x08080808, 0x08080404, 0x01010101]
    conns = []
    for c in range(12):
        dst = ext[c % 3] if c < 4 else rng.choice(svcs + pods + ext)
        conns.append(dict(pid=100 + c % 4, fd=3 + c, s=rng.choice(pods), d=dst, ce=hb.Encoder(), se=hb.Encoder(), next=1, tls=int(rng.random() < 0.3), last=0,
                          auth=[b"api.example.com", b"db.example.org:443", b"", b"api.example.com"][c % 4] if dst in ext else b"in-cluster"))
    recs = []; t = 10_000_000
    for _ in range(n_streams):
        c = rng.choice(conns); sid = c["next"]; c["next"] += 2
        grpc = rng.random() < 0.4
        req = c["ce"].block([(b":method", b"POST" if grpc else b"GET"), (b":scheme", b"http"), (b":path", rng.choice([b"/", b"/pkg.Svc/Call", b"/v1/items"]))] +
                            ([(b":authority", c["auth"])] if c["auth"] else []) + ([(b"content-type", b"application/grpc")] if grpc else []), huffman=rng.random() < 0.5)
        rsp = c["se"].block([(b":status", rng.choice([b"200", b"200", b"500", b"503"]))] + ([(b"grpc-status", rng.choice([b"0", b"13"]))] if grpc else []), huffman=rng.random() < 0.5)
        t += rng.randrange(1000, 90_000); lat = rng.randrange(10_000, 5_000_000)
        lat = max(lat, c["last"] + 1 - t); c["last"] = t + lat          # responses of one connection leave in encoder order
        recs.append((t, hb.l7_record(c["pid"], c["fd"], CLIENT, hb.frame(hb.HEADERS, sid, req), t, c["s"], c["d"], tls=c["tls"])))
        recs.append((t + lat, hb.l7_record(c["pid"], c["fd"], SERVER, hb.frame(hb.DATA, sid, b"..", flags=0) + hb.frame(hb.HEADERS, sid, rsp), t + lat, c["s"], c["d"], tls=c["tls"])))
        if rng.random() < 0.5:
            recs.append((t + 5, hb.l7_record(c["pid"], 99, 1, b"GET /user HTTP1.1\r\nHost: plain.example\r\n\r\n", t + 5, c["s"], rng.choice(svcs + ext), proto=1, status=200, dur=777)))
    recs.sort(key=lambda r: r[0])
    return b"".join(r for _, r in recs), sorted({c["pid"] for c in conns})


def test_http2_records_through_packer_equal_the_oracle_wire_path():
    topo = replay.make_topology(40, 200, seed=5)
    wire, pids = _h2_trace(topo, 600, seed=6)
    W = weights.make_weights(1)
    o = pyoracle.Oracle(*CLOCK, log_limit=5000); o.apply_ops(topo.k8s_ops())
    for p in pids[:-1]:                         # one pid is not live: its events are dropped on both sides
        o.h2().proc_exec(p)
    n_o = o.l7_wire(wire); o.window_close(W, 1)
    protos = {r[10] for r in o.reqinfos()}
    assert {"HTTP2", "HTTPS", "gRPC", "HTTP"} <= protos and o.h2().dropped_not_live() > 0

    pk = hostlib.Packer()
    for ip in list(topo.pod_ips) + list(topo.svc_ips):
        pk.known_ip(int(ip))
    for p in pids[:-1]:
        pk.proc_exec(p)
    packed = pk.pack_wire(wire)
    h2 = packed[packed["protocol"] == replay.PROTO_HTTP2]
    assert len(h2) > 300 and (h2["duration_ns"] >= 10_000).all()
    o2 = pyoracle.Oracle(*CLOCK); o2.apply_ops(topo.k8s_ops())
    n_p = o2.packed(packed, pk.labels); o2.window_close(W, 1)
    assert n_p == n_o and o2.edge_dict() == o.edge_dict() and o2.labels == o.labels == pk.labels
    assert b"api.example.com" in [l.encode() for l in pk.labels]


def test_graphds_ingest_wire_assembles_http2_and_tcp_close_drops_the_parser():
    cfg = engine.make_config(max_known_nodes=256, max_edges=4096)
    g = hostlib.GraphDS(cfg, engine_lib=None)
    g.PersistPod("p1", "10.0.0.1"); g.PersistService("s1", "10.96.0.1")
    A, S = 0x0A000001, 0x0A600001
    ce, se = hb.Encoder(), hb.Encoder()
    req = lambda sid: hb.frame(hb.HEADERS, sid, ce.block([(b":method", b"GET"), (b":path", b"/very/long/path/that/gets/indexed"), (b":authority", b"s1")]))
    rsp = lambda sid: hb.frame(hb.HEADERS, sid, se.block([(b":status", b"500")]))
    g.ingest_wire(hb.l7_record(50, 4, CLIENT, req(1), 1000, A, S))            # pid 50 not live yet: dropped
    g.proc_exec(50)
    ce = hb.Encoder()
    g.ingest_wire(hb.l7_record(50, 4, CLIENT, req(1), 1000, A, S) + hb.l7_record(50, 4, SERVER, rsp(1), 3500, A, S))
    st = g.http2_stats(); assert st["dropped_not_live"] == 1 and st["pending"] == 0 and st["parsers"] == 1
    # TCP close of (50, 4) — the line must exist (ESTABLISHED first) — drops the HPACK state: the next request of the
    # OLD encoder (indexed fields) cannot be decoded and is discarded
    def tcp(type_, ts):
        r = bytearray(64); r[0:8] = (4).to_bytes(8, "little"); r[8:16] = ts.to_bytes(8, "little"); r[16:20] = type_.to_bytes(4, "little"); r[20:24] = (50).to_bytes(4, "little")
        r[24:26] = (40000).to_bytes(2, "little"); r[26:28] = (8080).to_bytes(2, "little"); r[28:32] = bytes([10, 0, 0, 1]); r[44:48] = bytes([10, 96, 0, 1])
        return bytes(r)
    g.tcp_wire(tcp(1, 10) + tcp(5, 4000))
    assert g.http2_stats()["parsers"] == 0
    g.ingest_wire(hb.l7_record(50, 4, CLIENT, req(3), 5000, A, S) + hb.l7_record(50, 4, SERVER, rsp(3), 6000, A, S))
    assert g.http2_stats()["dropped_unparsed"] == 1
    g.FlushWindow()
    ev = g.mock_events()
    assert len(ev) == 1 and ev["protocol"][0] == replay.PROTO_HTTP2 and ev["duration_ns"][0] == 2500 and ev["status"][0] == 500 and ev["host_label"][0] == 0
