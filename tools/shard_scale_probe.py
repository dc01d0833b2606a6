#!/usr/bin/env python3
"""Per-shard kernel cost of the multi-GPU weak-scaling workload, measured on ONE GPU: WORLD logical shards
(one engine each, ThreadComm exchanges through device memory) run the workload of sharded.bench(); the
per-group HIP-event timings of shard 0's engine are what one GPU of a WORLD-GPU node spends in kernels
per window (exchanges excluded: those need the real fabric).  usage: shard_scale_probe.py [WORLD] [STEPS] [fixed|scaled] [CONFIG = 3]
(CONFIG 3 = BASELINE config 4: config 3's graph hash-sharded over WORLD GPUs, 10 M events per GPU and window)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alaz_amd import engine, sharded, replay, weights

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
gs = world if (len(sys.argv) > 3 and sys.argv[3] == "scaled") else 1      # graph: fixed (default) or scaled with world
cfgno = int(sys.argv[4]) if len(sys.argv) > 4 else 3
c = replay.CONFIGS[cfgno]; seed = replay.SEED_BASE + cfgno
Ev, L = c["events"], c["layers"]
topo = replay.make_topology(c["pods"] * gs, c["edges"] * gs, seed)
dev = torch.device("cuda", 0)
shared = sharded.ThreadComm.Shared(world)
engs, bes, evs = [], [], []
nlab = 64
for r in range(world):
    view = sharded.shard_view(topo, r, world)
    ev, labels = replay.make_events(view, Ev, seed + 7919 * (r + 1), fixed_labels=True)
    nlab = max(nlab, len(labels))
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(len(view.edge_src) * 1.25) + 4096, layers=L, max_labels=nlab,
                            max_outbound_ips=64, device=0, rank=r, world=world, max_batch=1 << 18, max_window_events=Ev)
    g.set_clock(1_000_000_000, 1_700_000_000_000_000_000); g.load_weights(weights.make_weights(L))
    for i in range(topo.n_pods): g.upsert_pod(int(topo.pod_ips[i]), i)
    for j in range(topo.n_svcs): g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
    g.set_label_count(len(labels))
    engs.append(g)
    evs.append(torch.from_numpy(ev.view(np.uint8).reshape(-1)).to(dev))
ncap = topo.n_nodes + nlab + 64
one = torch.cuda.Stream(dev)          # ONE stream for all shards: their kernels serialise, so the timings are uncontended
for r in range(world):
    bes.append(sharded.HipBackend(engs[r], ncap=ncap, layers=L, world=world, rank=r, device=dev, max_obip=64, stream=one))
torch.cuda.synchronize()

def worker(r, n):
    comm = sharded.ThreadComm(shared, r)
    for _ in range(n):
        engs[r].ingest_device(evs[r].data_ptr(), Ev, bes[r].s)
        sharded.run_window(bes[r], comm)
        engs[r].window_reset(bes[r].s)

def run(n):
    ths = [threading.Thread(target=worker, args=(r, n)) for r in range(world)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()

run(3)
engs[0].timing_reset(); engs[0].timing_enable(1)
t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
engs[0].timing_enable(0)
print(f"world {world}: N = {topo.n_nodes} known nodes, shard 0 edges ~{len(sharded.shard_view(topo, 0, world).edge_src)}, {Ev} events per shard and window")
names = {1: "K1a", 7: "K1b", 2: "K2", 8: "K3-in", 3: "K3-feat", 4: "K4", 5: "K5", 6: "K6-halo"}
per = {names[k]: round(engs[0].timing(k)[0] * engs[0].timing(k)[1] / steps, 1) for k in names}
print("shard 0 kernel groups, us per window:", per, " sum", round(sum(per.values()), 1))
print(f"all {world} shards on one GPU: {dt / steps * 1e6:.0f} us per step = {dt / steps / world * 1e6:.0f} us per shard-window")
