#!/usr/bin/env python
"""Host packer throughput (one core): 1096-byte l7_event records -> packed sg_event records, per protocol.
CPU only (no GPU needed).  The payload builders are the tests' (tests/h2_builder.py, tests/kafka_builder.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alaz_amd import build, hostlib, replay
from tests import h2_builder as hb, kafka_builder as kb

build.build_all()
topo = replay.make_topology(1000, 50_000, seed=1)
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(name, wire, setup=lambda pk: None, n_in=None):
    pk = hostlib.Packer(); setup(pk)
    for ip in list(topo.pod_ips) + list(topo.svc_ips):
        pk.known_ip(int(ip))
    n = len(wire) // 1096 if n_in is None else n_in
    pk.pack_wire(wire)                      # warm (statement caches, HPACK tables)
    t0 = time.perf_counter(); out = 0
    for _ in range(REP):
        out += len(pk.pack_wire(wire))
    dt = time.perf_counter() - t0
    print(f"{name:34s} {n * REP / dt / 1e6:7.2f} M records/s in   {out / dt / 1e6:7.2f} M events/s out   ({wire.__len__() * REP / dt / 1e9:5.2f} GB/s of records)")


ev, labels = replay.make_events(topo, 100_000, seed=2)
run("HTTP/1 (C2 mix, Host interning)", replay.to_wire(ev, labels))
ev, labels = replay.make_events(topo, 100_000, seed=3, mixed=True)
run("mixed protocols (pre-f-4 payloads)", replay.to_wire(ev, labels))

pods = [int(x) for x in topo.pod_ips[:64]]; svcs = [int(x) for x in topo.svc_ips[:16]]
recs = []; t = 1_000_000
encs = {}
for i in range(50_000):
    c = i % 256; ce, se = encs.setdefault(c, (hb.Encoder(), hb.Encoder()))
    sid = 1 + 2 * (i // 256)
    req = ce.block([(b":method", b"POST"), (b":scheme", b"http"), (b":path", b"/pkg.Service/Method%d" % (i % 7)), (b":authority", b"backend:8080"), (b"content-type", b"application/grpc")], huffman=True)
    rsp = se.block([(b":status", b"200"), (b"grpc-status", b"0")], huffman=True)
    recs.append(hb.l7_record(10, 3 + c, 1, hb.frame(hb.HEADERS, sid, req), t, pods[c % 64], svcs[c % 16])); t += 1000
    recs.append(hb.l7_record(10, 3 + c, 2, hb.frame(hb.HEADERS, sid, rsp), t, pods[c % 64], svcs[c % 16])); t += 1000
run("HTTP/2 gRPC (2 frames per request)", b"".join(recs), setup=lambda pk: pk.proc_exec(10))

batch = [kb.record(b"key-%d" % j, b"v" * 40, offset_delta=j) for j in range(8)]
for codec, nm in ((0, "none"), (2, "snappy"), (3, "lz4"), (4, "zstd"), (1, "gzip")):
    p = kb.produce_request([(b"orders", [(0, kb.record_batch(batch, codec=codec))])])
    f = kb.fetch_response([(b"orders", [(0, kb.record_batch(batch, codec=codec))])])
    wire = b"".join(kb.l7_record(1 + (i & 1), f if i & 1 else p, 1000 + i, pods[i % 64], svcs[i % 16], api_version=11 if i & 1 else 7) for i in range(20_000))
    run(f"Kafka, 8 records/batch, {nm}", wire, setup=lambda pk: pk.kafka_decode(True))
