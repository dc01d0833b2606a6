"""Experiment: N engine handles alternating windows on N streams (windows in flight overlap)."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from alaz_amd import engine, replay, weights
ne = int(sys.argv[1]) if len(sys.argv) > 1 else 2
c = replay.CONFIGS[2]; seed = replay.SEED_BASE + 2; Ev, L = c["events"], c["layers"]; nb = 11
topo = replay.make_topology(c["pods"], c["edges"], seed)
ev_all, labels = replay.make_events(topo, Ev * nb, seed)
engs, streams = [], []
for k in range(ne):
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(c["edges"] * 1.25) + 4096, layers=L, max_labels=max(64, len(labels)),
                            max_outbound_ips=64, max_batch=1 << 18, max_window_events=Ev)
    g.set_clock(1_000_000_000, 1_700_000_000_000_000_000); g.load_weights(weights.make_weights(L))
    for i in range(topo.n_pods): g.upsert_pod(int(topo.pod_ips[i]), i)
    for j in range(topo.n_svcs): g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
    g.set_label_count(len(labels)); engs.append(g); streams.append(torch.cuda.Stream())
dev = [torch.from_numpy(ev_all[i * Ev:(i + 1) * Ev].view(np.uint8).reshape(-1)).cuda() for i in range(nb)]
torch.cuda.synchronize()
def step(i):
    k = i % ne; s = streams[k].cuda_stream
    engs[k].ingest_device(dev[i % nb].data_ptr(), Ev, s); engs[k].window_run(s)
for i in range(20): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 200
for i in range(K): step(20 + i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"engines": ne, "ms_per_step": dt / K * 1e3, "events_per_s": Ev * K / dt}))
