#!/usr/bin/env python3
"""One shard of eight of BASELINE config 5, the engine exactly as `bench.py --config 5 --shard-of 8` builds it (sg_create's own K1 rule,
no kept state), one window against the CPU oracle row for row — and the per-kernel durations of a few windows (rocprofv3 wraps this)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alaz_amd import engine, replay, sharded, weights
from oracle import pyoracle
from tests.helpers import CLOCK, HostShim, compare_edge_dicts, engine_edge_dict

c = replay.CONFIGS[5]; seed = replay.SEED_BASE + 5
full = replay.make_topology(c["pods"], c["edges"], seed)
topo = sharded.shard_view(full, 0, 8)
Ev = c["events"] // 8
ev, labels = replay.make_events(topo, Ev, seed, mixed=True)
L = c["layers"]
n_edges = min(len(topo.edge_src), Ev)
g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(n_edges * 1.25) + 4096, layers=L, max_labels=max(64, len(labels)),
                        max_outbound_ips=64, max_batch=1 << 20, max_window_events=Ev, warm=False)
print("geometry", g.geometry(), flush=True)
g.set_clock(*CLOCK); W = weights.make_weights(L); g.load_weights(W)
shim = HostShim(); shim.apply(g, topo.k8s_ops())
g.set_label_count(len(labels))
dev = torch.from_numpy(ev.view(np.uint8).reshape(-1)).cuda()
for k in range(4):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g.ingest_device(dev.data_ptr(), len(ev), 0); g.window_run(0)
    torch.cuda.synchronize(); print(f"window {k}: {(time.perf_counter() - t1) * 1e3:.3f} ms", flush=True)
g.ingest_device(dev.data_ptr(), len(ev), 0)
rows = g.flush_window()
o = pyoracle.Oracle(*CLOCK); o.apply_ops(topo.k8s_ops()); o.packed(ev, labels); o.window_close(W, L)
worst = compare_edge_dicts(engine_edge_dict(rows, shim, labels, g.outbound_ips()), o.edge_dict())
orow = o.edge_rows()
assert np.array_equal(rows["from_ref"], orow["from_ref"]) and np.array_equal(rows["to_ref"], orow["to_ref"])
print(f"C5 shard of 8: {len(rows)} rows equal the oracle's, max |score diff| {worst:.2e}")
