import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alaz_amd import engine, replay, weights
c = replay.CONFIGS[3]; seed = replay.SEED_BASE + 3
topo = replay.make_topology(c["pods"], c["edges"], seed)
ev, labels = replay.make_events(topo, 10_000_000, seed)
for n in (3_000_000, 10_000_000):
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(c["edges"] * 1.25) + 4096, layers=2, max_labels=64, max_outbound_ips=64, max_batch=1 << 18, max_window_events=10_000_000)
    g.set_clock(1_000_000_000, 1_700_000_000_000_000_000); g.load_weights(weights.make_weights(2))
    for i in range(topo.n_pods): g.upsert_pod(int(topo.pod_ips[i]), i)
    for j in range(topo.n_svcs): g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
    g.set_label_count(len(labels))
    d = torch.from_numpy(ev[:n].view(np.uint8).reshape(-1)).cuda()
    g.ingest_device(d.data_ptr(), n, 0); torch.cuda.synchronize(); print(n, "ingested", flush=True)
    g.window_run(0); torch.cuda.synchronize()
    rows = g.window_read(); st = g.stats()
    print(n, "rows", len(rows), "events", st.last_window_events, "dropcap", st.events_dropped_cap, int(rows["count"].sum()), g.geometry()["pass_a_teams"], flush=True)
    g.close()
