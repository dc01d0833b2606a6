#!/usr/bin/env python3
"""K1 tuning sweep on one GPU box: for each environment variant (SG_NP, SG_HT, SG_CT, SG_K1B_THREADS, SG_K1B_U, SG_ABLATE, ...)
create an engine, run a few windows of one BASELINE config from device-resident batches, print the K1a / K1b kernel
durations (dispatch stamps), the window time and the drop counters.  Usage: k1_sweep.py CONFIG 'A=1 B=2' 'A=3' ...
The trace is generated once and cached under /tmp (the box is fresh per gpurun call, so once per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from alaz_amd import engine, replay, weights

cfgno = int(sys.argv[1]); variants = sys.argv[2:] or [""]
c = replay.CONFIGS[cfgno]; seed = replay.SEED_BASE + cfgno
Ev, L = c["events"], c["layers"]
nb = 2 if cfgno == 3 else 6
topo = replay.make_topology(c["pods"], c["edges"], seed)
cache = f"/tmp/sweep_ev_c{cfgno}.npy"
if os.path.exists(cache):
    ev = np.load(cache); labels = ["x"] * 64
else:
    ev, labels = replay.make_events(topo, Ev * nb, seed); np.save(cache, ev)
dev = [torch.from_numpy(ev[i * Ev:(i + 1) * Ev].view(np.uint8).reshape(-1)).cuda() for i in range(nb)]
torch.cuda.synchronize()
W = weights.make_weights(L)
steps = int(os.environ.get("SWEEP_STEPS", "8" if cfgno == 3 else "40"))
base_env = dict(os.environ)
for var in variants:
    os.environ.clear(); os.environ.update(base_env)
    for kv in var.split():
        k, v = kv.split("="); os.environ[k] = v
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(c["edges"] * 1.25) + 4096, layers=L, max_labels=64,
                            max_outbound_ips=64, max_batch=1 << 18, max_window_events=Ev,
                            edge_histogram=bool(int(os.environ.get("SG_SWEEP_HIST", "0"))))
    g.set_clock(1_000_000_000, 1_700_000_000_000_000_000); g.load_weights(W)
    for i in range(topo.n_pods): g.upsert_pod(int(topo.pod_ips[i]), i)
    for j in range(topo.n_svcs): g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
    g.set_label_count(64)
    for i in range(3):
        g.ingest_device(dev[i % nb].data_ptr(), Ev, 0); g.window_run(0)
    torch.cuda.synchronize()
    g.timing_reset(); g.timing_enable((1 << 1) | (1 << 7))
    t0 = time.perf_counter()
    for i in range(steps):
        g.ingest_device(dev[i % nb].data_ptr(), Ev, 0); g.window_run(0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e6
    g.timing_enable(0)
    a = g.timing(1)[0]; b = g.timing(7)[0] * g.timing(7)[1] / steps      # (a warm engine's pass B is two launches per window: the sum)
    g.timing_reset(); g.timing_enable(1)
    for i in range(4):
        g.ingest_device(dev[i % nb].data_ptr(), Ev, 0); g.window_run(0)
    torch.cuda.synchronize(); g.timing_enable(0)
    grp = {k: round(g.timing(k)[0] * g.timing(k)[1] / 4, 1) for k in (2, 8, 3, 4, 5)}      # us per window (a group may have several records)
    g.ingest_device(dev[0].data_ptr(), Ev, 0); torch.cuda.synchronize()
    rows = g.flush_window(); st = g.stats()
    E = int(st.last_window_edges)
    frac = (32.0 * Ev + 32.0 * E) / ((a + b) * 1e-6) / 8e12 if a + b > 0 else 0
    geo = g.geometry()
    print(f"[{var or 'default'}] {'narrow' if geo['k1_narrow'] else 'wide'} np {geo['partitions']} x{geo['pass_b_split']} ht {geo['table_slots']} ct {geo['cache_slots']} l2lds {geo['join_l2_in_lds']} | k1a {a:7.1f} k1b {b:7.1f} us  frac {frac:.3f}  window {dt:8.1f} us  groups {grp}  edges {E} events {st.last_window_events} "
          f"dropped_cap {st.events_dropped_cap} count_sum {int(rows['count'].astype(np.uint64).sum())}", flush=True)
    g.close()
