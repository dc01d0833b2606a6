#!/usr/bin/env python3
"""BASELINE config 5 at full size on one GPU (100k pods / 50k services / 20M edges, 5M mixed-protocol events per
window, L = 2): does the engine hold it (variant selection, 8 GB of window buffers), and do the size-independent
invariants hold (count conservation, canonical row order, every edge's endpoints valid)?  Prints per-window time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alaz_amd import engine, replay, weights

c = replay.CONFIGS[5]; seed = replay.SEED_BASE + 5
t0 = time.time(); topo = replay.make_topology(c["pods"], c["edges"], seed)
ev, labels = replay.make_events(topo, c["events"], seed, mixed=True); print(f"generated in {time.time() - t0:.0f} s", flush=True)
L = c["layers"]
g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(c["edges"] * 1.1), layers=L, max_labels=max(64, len(labels)),
                        max_outbound_ips=64, max_batch=1 << 20, max_window_events=len(ev))
g.set_clock(1_000_000_000, 1_700_000_000_000_000_000); g.load_weights(weights.make_weights(L))
for i in range(topo.n_pods): g.upsert_pod(int(topo.pod_ips[i]), i)
for j in range(topo.n_svcs): g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
g.set_label_count(len(labels))
dev = torch.from_numpy(ev.view(np.uint8).reshape(-1)).cuda()
for k in range(3):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g.ingest_device(dev.data_ptr(), len(ev), 0); g.window_run(0)
    torch.cuda.synchronize(); print(f"window {k}: {(time.perf_counter() - t1) * 1e3:.2f} ms", flush=True)
g.ingest_device(dev.data_ptr(), len(ev), 0)
rows = g.flush_window()
st = g.stats()
known = np.isin(ev["saddr"], topo.pod_ips)
print("edges", len(rows), "events accepted", st.last_window_events, "dropped src", st.events_dropped_src, "cap", st.events_dropped_cap, "nodes", st.last_window_nodes)
assert int(rows["count"].sum()) == st.last_window_events == int(known.sum()) and st.events_dropped_cap == 0
key = (rows["from_ref"].astype(np.uint64) << np.uint64(32)) | rows["to_ref"].astype(np.uint64)
assert np.all(key[1:] > key[:-1]), "rows not in canonical order / duplicates"
assert np.isfinite(rows["score"]).all() and (rows["score"] > 0).all() and (rows["score"] < 1).all()
print("C5 invariants OK")
