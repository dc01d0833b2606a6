"""SURVEY.md §8 f-4 (second half): Kafka payload decode on the host.

* Checksums and decompressors of both implementations — the oracle's restatement (oracle/kafka.c) and the
  product's host side (alaz_amd/csrc/host/kafka.cpp) — against published check values and against data
  compressed by independent implementations (pyarrow codecs, zlib, the xxhash module).
* The reference's decode rules (aggregator/data.go:929-1017 + aggregator/kafka/*.go) as scenario tests on both.
* C++ vs oracle differentially on random, mutated and truncated payloads.
* Kafka records through the packer -> packed events -> same edges as the oracle's wire path.
The reference holds no test for this path (parity unpinned by reference tests).  All CPU-only."""
import gzip
import random
import struct
import zlib

import numpy as np
import pyarrow as pa
import pytest
import xxhash

from alaz_amd import build, hostlib, replay, weights
from oracle import pyoracle
from tests import kafka_builder as kb

CLOCK = (1_000_000_000, 1_700_000_000_000_000_000)
PRODUCE, FETCH = 1, 2


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build_all()


IMPL = [("oracle", pyoracle), ("host", hostlib)]


def both(payload, method, version=0):
    a, b = pyoracle.kafka_decode(payload, method, version), hostlib.kafka_decode(payload, method, version)
    assert a == b, (a, b)
    return a


# ------------------------------------------------------------------------------------------------ checksums / codecs
@pytest.mark.parametrize("who,m", IMPL)
def test_checksum_known_answers(who, m):
    assert m.crc32(b"123456789") == 0xCBF43926 and m.crc32(b"123456789", True) == 0xE3069283     # the CRC catalogue's check values
    assert m.crc32(b"", True) == 0 and m.xxh32(b"") == 0x02CC5D05
    rng = random.Random(1)
    for n in list(range(0, 40)) + [255, 256, 1000]:
        d = bytes(rng.randrange(256) for _ in range(n))
        assert m.crc32(d) == zlib.crc32(d) and m.crc32(d, True) == kb.crc32c(d)
        for seed in (0, 1, 0x9E3779B1):
            assert m.xxh32(d, seed) == xxhash.xxh32(d, seed=seed).intdigest()


SAMPLES = [b"", b"a", b"hello kafka " * 40, bytes(range(256)) * 9, b"\x00" * 70000, bytes(random.Random(3).randrange(256) for _ in range(5000))]


@pytest.mark.parametrize("who,m", IMPL)
@pytest.mark.parametrize("codec,name", [(1, "gzip"), (2, "snappy"), (3, "lz4"), (4, "zstd")])
def test_decompressors_against_independent_compressors(who, m, codec, name):
    for d in SAMPLES:
        c = pa.Codec(name).compress(d, asbytes=True)
        # go-xerial-snappy refuses inputs shorter than its 8-byte magic (the empty block is 1 byte)
        assert m.kafka_decompress(codec, c) == (None if name == "snappy" and len(c) < 8 else d)
    assert m.kafka_decompress(0, b"as is") == b"as is"
    assert m.kafka_decompress(5, b"x") is None and m.kafka_decompress(7, b"") is None       # "invalid compression specified"


@pytest.mark.parametrize("who,m", IMPL)
def test_decompressor_framings_and_rejections(who, m):
    d = SAMPLES[2]
    # gzip: concatenated members are one stream; a cut member, trailing garbage and an empty input are errors
    assert m.kafka_decompress(1, gzip.compress(d) + gzip.compress(b"tail")) == d + b"tail"
    assert m.kafka_decompress(1, gzip.compress(d)[:-3]) is None and m.kafka_decompress(1, gzip.compress(d) + b"junk") is None and m.kafka_decompress(1, b"") is None
    # snappy: xerial framing (two chunks) and the bare block; short input, cut chunk, bad offsets are errors
    assert m.kafka_decompress(2, kb.compress(kb.SNAPPY, d, xerial=True)) == d
    assert m.kafka_decompress(2, kb.XERIAL_HEADER + struct.pack(">I", 100) + b"xx") is None and m.kafka_decompress(2, b"short") is None
    assert m.kafka_decompress(2, bytes([5, 0x01, 0x00]) + b"\x00" * 6) is None                 # copy before any output
    raw = pa.Codec("snappy").compress(d, asbytes=True)
    assert m.kafka_decompress(2, raw[:-1]) is None
    # lz4: a hand-made frame of stored blocks with block and content checksums, a skippable frame in front, two frames in a row
    def frame(data, flip=None):
        flg = 0x40 | 0x20 | 0x10 | 0x04; bd = 0x40                                            # v01, independent, block checksum, content checksum; 64 KiB blocks
        hdr = bytes([flg, bd]); out = struct.pack("<I", 0x184D2204) + hdr + bytes([(xxhash.xxh32(hdr).intdigest() >> 8) & 0xFF])
        for i in range(0, len(data), 65536):
            blk = data[i:i + 65536]
            out += struct.pack("<I", len(blk) | 0x80000000) + blk + struct.pack("<I", xxhash.xxh32(blk).intdigest())
        out += struct.pack("<I", 0) + struct.pack("<I", xxhash.xxh32(data).intdigest())
        if flip is not None:
            out = out[:flip] + bytes([out[flip] ^ 1]) + out[flip + 1:]
        return out
    big = SAMPLES[4] + SAMPLES[5]
    skippable = struct.pack("<II", 0x184D2A53, 5) + b"meta!"
    assert m.kafka_decompress(3, skippable + frame(big) + frame(b"second")) == big + b"second"
    assert m.kafka_decompress(3, frame(big, flip=6)) is None                                   # header checksum
    assert m.kafka_decompress(3, frame(big, flip=20)) is None                                  # block payload -> block checksum
    assert m.kafka_decompress(3, frame(big)[:-1]) is None and m.kafka_decompress(3, b"\x04\x22\x4d") is None
    c = pa.Codec("lz4").compress(d, asbytes=True)
    assert m.kafka_decompress(3, c[:-2]) is None
    # zstd: two frames in a row; a cut frame is an error
    z = pa.Codec("zstd").compress(d, asbytes=True)
    assert m.kafka_decompress(4, z + pa.Codec("zstd").compress(b"more", asbytes=True)) == d + b"more" and m.kafka_decompress(4, z[:-2]) is None


# ------------------------------------------------------------------------------------------------ protocol scenarios
RECS = [kb.record(b"k1", b"v1"), kb.record(None, b"value-two", offset_delta=1, headers=[(b"trace", b"abc"), (b"n", None)]), kb.record(b"k3", None, offset_delta=2)]
MSGS = [(b"k1", b"v1"), (b"", b"value-two"), (b"k3", b"")]


def test_produce_request_versions_codecs_and_order():
    for version in (3, 4, 5, 6, 7):
        for codec in range(5):
            p = kb.produce_request([(b"orders", [(3, kb.record_batch(RECS[:1], codec=codec)), (0, kb.record_batch(RECS, codec=codec))]),
                                    (b"audit", [(1, kb.record_batch(RECS[1:], codec=codec, xerial=True))])], version=version,
                                   transactional_id=b"txn" if version % 2 else None)
            st, msgs = both(p, PRODUCE)
            assert st == "ok" and msgs == [(b"audit", 1) + m for m in MSGS[1:]] + [(b"orders", 0) + m for m in MSGS] + [(b"orders", 3) + MSGS[0]]
    # versions 0..2 carry no transactional id; their (legacy) message sets end in the reference's nil-RecordBatch panic
    assert both(kb.produce_request([(b"t", [(0, kb.legacy_message(b"k", b"v"))])], version=2), PRODUCE) == ("panic", [])
    assert both(kb.produce_request([(b"t", [(0, kb.legacy_message(b"k", b"v", magic=0) + kb.legacy_message(None, b"w", offset=1))])], version=0), PRODUCE) == ("panic", [])
    # ... but a record batch inside a v2 request decodes (the magic byte decides, not the api version)
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch(RECS))])], version=2), PRODUCE)[0] == "ok"
    # no topics: decodes to nothing (the event is then dropped for having no message)
    assert both(kb.produce_request([]), PRODUCE) == ("ok", [])


def test_produce_request_go_map_semantics():
    """Records is map[string]map[int32]Records: a repeated topic starts over, a repeated partition is overwritten."""
    a, b = kb.record_batch(RECS[:1]), kb.record_batch(RECS[1:])
    assert both(kb.produce_request([(b"t", [(0, a), (0, b)])]), PRODUCE) == ("ok", [(b"t", 0) + m for m in MSGS[1:]])
    assert both(kb.produce_request([(b"t", [(0, a), (1, a)]), (b"u", [(5, a)]), (b"t", [(2, b)])]), PRODUCE) == \
        ("ok", [(b"t", 2) + m for m in MSGS[1:]] + [(b"u", 5) + MSGS[0]])
    # a legacy set that is overwritten by a later record batch no longer panics
    assert both(kb.produce_request([(b"t", [(0, kb.legacy_message(b"k", b"v")), (0, a)])]), PRODUCE) == ("ok", [(b"t", 0) + MSGS[0]])
    # a null topic name reads as ""
    assert both(kb.produce_request([(None, [(0, a)])]), PRODUCE) == ("ok", [(b"", 0) + MSGS[0]])


def test_produce_request_rejections():
    ok = kb.produce_request([(b"t", [(0, kb.record_batch(RECS))])])
    assert both(ok, PRODUCE)[0] == "ok"
    for bad in (ok[:-1], ok[:20], ok[:3], b"",                                   # body shorter than the size prefix says (1 KiB capture): unexpected EOF
                kb.produce_request([(b"t", [(0, kb.record_batch(RECS))])], size_delta=-1),      # body longer than decoded: the tail belongs to nothing
                kb.produce_request([(b"t", [(0, kb.record_batch(RECS))])], trailing=b"\x00"),   # "invalid length"
                kb.produce_request([(b"t", [(0, kb.record_batch(RECS))])], api_key=1),          # only Produce is a known request
                kb.produce_request([(b"t", [(0, kb.record_batch(RECS, bad_crc=True))])]),
                kb.produce_request([(b"t", [(0, kb.record_batch(RECS, codec=5))])]),            # unknown codec
                kb.produce_request([(b"t", [(0, kb.record_batch(RECS, codec=kb.GZIP)[:-4] + b"\0\0\0\0")])]),
                struct.pack(">i", 3) + b"abc", struct.pack(">i", 200 * 1024 * 1024) + b"x" * 64):
        st, msgs = both(bad, PRODUCE)
        assert st in ("error", "insufficient") and msgs == []
    # declared fewer records than present: the rest is "invalid length"; declared more: a partial trailing record => no message, no error
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch(RECS, num_records=2))])]), PRODUCE) == ("error", [])
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch(RECS, num_records=4))]), (b"u", [(0, kb.record_batch(RECS[:1]))])]), PRODUCE) == ("ok", [(b"u", 0) + MSGS[0]])
    # batch length pointing past the records bytes: PartialTrailingRecord, Records = nil
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch(RECS, batch_len_delta=9))]), (b"u", [(0, kb.record_batch(RECS[:1]))])]), PRODUCE) == ("ok", [(b"u", 0) + MSGS[0]])
    # batch length below the 49-byte overhead: negative slice length
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch([], batch_len_delta=-10))])]), PRODUCE)[0] == "error"
    # a record whose length varint is not minimal fails varintLengthField.check
    body_len = len(RECS[0]) - 1
    fat = kb.record(b"k1", b"v1", length_override=bytes([(body_len << 1) | 0x80, 0x00]))
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch([fat]))])]), PRODUCE) == ("error", [])
    # control / transactional attribute bits do not matter
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch(RECS, attributes_extra=0x30))])]), PRODUCE)[0] == "ok"
    # a snappy block of length 0 (only reachable with an over-long varint: the xerial wrapper wants >= 8 bytes): golang/snappy
    # hands back a nil slice, decode(nil) returns at once and the nil *Record entries blow up in decodeKafkaPayload
    empty = bytes([0x80] * 7 + [0x00])
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch([], codec=kb.SNAPPY, num_records=1, raw_payload=empty))])]), PRODUCE) == ("panic", [])
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch([], codec=kb.SNAPPY, num_records=0, raw_payload=empty))])]), PRODUCE) == ("ok", [])
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch([], codec=kb.SNAPPY, num_records=0))])]), PRODUCE) == ("error", [])


def test_fetch_response_versions_and_record_sets():
    two = kb.record_batch(RECS) + kb.record_batch(RECS[:2], base_offset=3, codec=kb.LZ4)
    for version in range(0, 12):
        p = kb.fetch_response([(b"orders", [(0, two), (2, kb.record_batch(RECS[2:]))])], version=version, aborted=2 if version >= 4 else 0)
        assert both(p, FETCH, version) == ("ok", [(b"orders", 0) + m for m in MSGS + MSGS[:2]] + [(b"orders", 2) + MSGS[2]]), version
    # decoding a v11 body as v3 (wrong api version from the kernel side) misreads it: garbage or error, identically on both sides
    both(kb.fetch_response([(b"orders", [(0, two)])], version=11), FETCH, 3)
    # api version >= 12: header v1 with tagged fields, then a non-flexible body decode — as the reference does
    both(kb.fetch_response([(b"orders", [(0, two)])], version=11), FETCH, 12)
    # a trailing batch cut by the fetch size is dropped silently; a first batch that is cut yields nothing
    cut = kb.record_batch(RECS) + kb.record_batch(RECS[:2], base_offset=3)[:40]
    assert both(kb.fetch_response([(b"t", [(0, cut)])]), FETCH, 11) == ("ok", [(b"t", 0) + m for m in MSGS])
    assert both(kb.fetch_response([(b"t", [(0, kb.record_batch(RECS)[:30])])]), FETCH, 11) == ("ok", [])
    assert both(kb.fetch_response([(b"t", [(0, kb.record_batch(RECS, batch_len_delta=50) )])]), FETCH, 11) == ("ok", [])
    # empty record set, several topics, header length field out of range, records cut by the capture
    assert both(kb.fetch_response([(b"t", [(0, b"")]), (b"u", [(1, kb.record_batch(RECS[:1]))])]), FETCH, 11) == ("ok", [(b"u", 1) + MSGS[0]])
    assert both(kb.fetch_response([(b"t", [(0, two)])], size=4), FETCH, 11)[0] == "error"
    assert both(kb.fetch_response([(b"t", [(0, two)])])[:-5], FETCH, 11)[0] == "insufficient"
    # a legacy message set in a fetch response with at least one message: nil RecordBatch => panic => nothing
    assert both(kb.fetch_response([(b"t", [(0, kb.legacy_message(b"k", b"v"))])], version=3), FETCH, 3) == ("panic", [])
    assert both(kb.fetch_response([(b"t", [(0, kb.legacy_message(b"k", kb.compress(kb.GZIP, kb.legacy_message(b"a", b"b")), codec=kb.GZIP))])], version=3), FETCH, 3) == ("panic", [])
    # bad CRC in a legacy message is an error before that
    bad = bytearray(kb.legacy_message(b"k", b"v")); bad[-1] ^= 1
    assert both(kb.fetch_response([(b"t", [(0, bytes(bad))])], version=3), FETCH, 3) == ("error", [])
    # neither method: nothing, no error
    assert both(kb.produce_request([(b"t", [(0, kb.record_batch(RECS))])]), 3) == ("ok", [])


def test_negative_counts_pass_get_array_length():
    """getArrayLength lets negative counts through (real_decoder.go:113-127); loops then simply do not run."""
    p = bytearray(kb.produce_request([(b"t", [(0, kb.record_batch(RECS))])]))
    hdr = 4 + 8 + 2 + len(b"producer-1") + 2 + 6                    # size, key/version/correlation, client id, txn id (-1), acks+timeout
    assert struct.unpack_from(">i", p, hdr)[0] == 1
    struct.pack_into(">i", p, hdr, -1)
    assert both(bytes(p), PRODUCE) == ("error", [])                 # nothing decoded, the body is left over: "invalid length"
    f = bytearray(kb.fetch_response([(b"t", [(0, kb.record_batch(RECS))])], version=0))
    struct.pack_into(">i", f, 8, -1)
    assert both(bytes(f), FETCH, 0) == ("ok", [])


def test_differential_random_and_mutated_payloads():
    rng = random.Random(4242)
    def rand_bytes(lo, hi): return bytes(rng.randrange(256) for _ in range(rng.randrange(lo, hi)))
    def rand_records():
        return [kb.record(rand_bytes(0, 6) if rng.random() < 0.8 else None, rand_bytes(0, 30) if rng.random() < 0.9 else None, offset_delta=i,
                          headers=[(rand_bytes(1, 4), rand_bytes(0, 4)) for _ in range(rng.randrange(0, 3))]) for i in range(rng.randrange(0, 5))]
    agree = ok = 0
    for trial in range(600):
        topics = []
        for t in range(rng.randrange(0, 3)):
            parts = []
            for _ in range(rng.randrange(0, 3)):
                r = rng.random()
                if r < 0.1:
                    recs = kb.legacy_message(rand_bytes(0, 4), rand_bytes(0, 9), magic=rng.randrange(2))
                else:
                    recs = b"".join(kb.record_batch(rand_records(), codec=rng.choice([0, 0, 1, 2, 3, 4]), xerial=rng.random() < 0.5, base_offset=i * 10) for i in range(rng.randrange(1, 3)))
                parts.append((rng.randrange(0, 3), recs))
            topics.append((rng.choice([b"a", b"b", b"topic-long-name"]), parts))
        if rng.random() < 0.5:
            method, version = PRODUCE, rng.choice([3, 5, 7])
            topics = [(n, [(pid, rs if rs[16:17] < b"\x02" else rs) for pid, rs in ps]) for n, ps in topics]
            p = kb.produce_request(topics, version=version)
        else:
            method, version = FETCH, rng.randrange(0, 12)
            p = kb.fetch_response(topics, version=version, aborted=rng.randrange(0, 2) if version >= 4 else 0)
        r = rng.random()
        if r < 0.35 and p:
            i = rng.randrange(len(p)); p = p[:i] + bytes([p[i] ^ (1 << rng.randrange(8))]) + p[i + 1:]
        elif r < 0.5:
            p = p[:rng.randrange(0, len(p) + 1)]
        elif r < 0.55:
            p = p + rand_bytes(1, 5)
        p = p[:1024]
        st, msgs = both(p, method, version)
        agree += 1; ok += st == "ok" and bool(msgs)
    assert ok > 100
    for _ in range(2000):                                             # pure noise
        p = rand_bytes(0, 80)
        both(p, rng.choice([PRODUCE, FETCH]), rng.randrange(0, 14))


# ------------------------------------------------------------------------------------------------ through the packer
def test_kafka_records_through_packer_equal_the_oracle_wire_path():
    topo = replay.make_topology(30, 120, seed=21)
    rng = random.Random(22)
    pods = [int(x) for x in topo.pod_ips[:12]]; brokers = [int(x) for x in topo.svc_ips[:3]] + [0x0A0A0A0A]      # a service IP ... and an outbound broker
    wire = []; t = 5_000_000; want_msgs = 0; want_drops = 0
    for i in range(400):
        s, d = rng.choice(pods), rng.choice(brokers)
        n = rng.randrange(1, 5); recs = [kb.record(b"k%d" % j, b"v" * rng.randrange(1, 20), offset_delta=j) for j in range(n)]
        codec = rng.choice([0, 1, 2, 3, 4]); kind = rng.random()
        if kind < 0.45:
            p, m, ver = kb.produce_request([(b"orders", [(i % 3, kb.record_batch(recs, codec=codec))])], version=7), PRODUCE, 7
        elif kind < 0.85:
            p, m, ver = kb.fetch_response([(b"orders", [(i % 3, kb.record_batch(recs, codec=codec))])], version=11), FETCH, 11
        elif kind < 0.9:
            p, m, ver = kb.produce_request([(b"big", [(0, kb.record_batch([kb.record(b"k", b"x" * 2000)]))])], version=7), PRODUCE, 7; n = 0      # > 1 KiB capture
        elif kind < 0.95:
            p, m, ver = kb.produce_request([(b"old", [(0, kb.legacy_message(b"k", b"v"))])], version=2), PRODUCE, 2; n = 0
        else:
            p, m, ver = b"\x00\x00\x00\x10not kafka at all", PRODUCE, 0; n = 0
        want_msgs += n; want_drops += n == 0
        t += rng.randrange(1000, 50_000)
        wire.append(kb.l7_record(m, p, t, s, d, api_version=ver, dur=rng.randrange(1000, 900_000)))
    wire = b"".join(wire)
    W = weights.make_weights(1)
    o = pyoracle.Oracle(*CLOCK, log_limit=5000); o.apply_ops(topo.k8s_ops()); o.set_kafka_decode(True)
    assert o.l7_wire(wire) == want_msgs and o.dropped_parse == want_drops
    o.window_close(W, 1)
    rows = o.reqinfos()
    assert {r[13] for r in rows} == {"PUBLISH", "CONSUME"} and {r[14] for r in rows} == {"orders"}

    pk = hostlib.Packer(); pk.kafka_decode(True)
    for ip in list(topo.pod_ips) + list(topo.svc_ips):
        pk.known_ip(int(ip))
    packed = pk.pack_wire(wire)
    assert len(packed) == want_msgs and pk.dropped_parse == want_drops
    assert ((packed["flags"] & replay.EV_CONSUME) != 0).sum() == sum(1 for r in rows if r[13] == "CONSUME")
    o2 = pyoracle.Oracle(*CLOCK); o2.apply_ops(topo.k8s_ops())
    assert o2.packed(packed, pk.labels) == want_msgs
    o2.window_close(W, 1)
    assert o2.edge_dict() == o.edge_dict()
