#!/usr/bin/env python3
"""What a window costs when it keeps meeting edges the kept set lacks (warm attempt -> unknown key -> rebuild with the kept set grown):
BASELINE config 3's graph, window k draws its 10 M events from the first 600 k + 40 k x k edges, so every window brings ~40 k new
edges.  The host's policy (servicegraph.hip do_close: the device's note, three attempts in a row that met unknown keys -> 32 plain closes)
is what the first line shows at work.  Prints per-window GPU time (one hipEvent pair per window) for that stream, for the same stream with the warm path switched off,
and for a steady stream (same trace every window)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alaz_amd import engine, replay, weights

c = replay.CONFIGS[3]; seed = replay.SEED_BASE + 3
topo = replay.make_topology(c["pods"], c["edges"], seed)
Ev, W = 4_000_000, 10
def sub(n_edges, s):
    t = replay.Topology(topo.n_pods, topo.n_svcs, topo.pod_ips, topo.svc_ips, topo.edge_src[:n_edges], topo.edge_dst[:n_edges], topo.seed)
    return replay.make_events(t, Ev, s, fixed_labels=True)[0]
traces = [sub(600_000 + 40_000 * k, seed + k) for k in range(W)]
dev = [torch.from_numpy(t.view(np.uint8).reshape(-1)).cuda() for t in traces]
def run(warm, steady):
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=1_250_000, layers=2, max_labels=128, max_outbound_ips=64, max_window_events=Ev,
                            warm=warm)
    g.set_clock(1_000_000_000, 1_700_000_000_000_000_000); g.load_weights(weights.make_weights(2))
    for i in range(topo.n_pods): g.upsert_pod(int(topo.pod_ips[i]), i)
    for j in range(topo.n_svcs): g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
    g.set_label_count(128)
    for _ in range(2):
        g.ingest_device(dev[0].data_ptr(), Ev, 0); g.window_run(0)
    torch.cuda.synchronize()
    g.timing_reset(); g.timing_enable(1 << 10)
    g.window_read(); st = g.stats(); seen = (st.windows_warm, st.windows_cold, st.windows_plain, st.windows_delta); paths = []
    for k in range(W):
        g.ingest_device(dev[0 if steady else k].data_ptr(), Ev, 0); g.window_run(0)
        torch.cuda.synchronize()                             # windows are seconds apart in production: the host closes one knowing how the last one went
        g.window_read(); st = g.stats()                      # (untimed: group 10 ends behind the score kernel) — which path the window took
        paths.append("cold" if st.windows_cold > seen[1] else ("plain" if st.windows_plain > seen[2] else (f"delta+{st.last_window_new_edges}" if st.windows_delta > seen[3] else "warm")))
        seen = (st.windows_warm, st.windows_cold, st.windows_plain, st.windows_delta)
        if os.environ.get("SG_ABLATE") and k in (0, 3):          # phase stamps of this window's kw_compact (SG_ABLATE=0x100)
            a = g.debug_stamps()[2].astype(np.int64); a = a[a[:, 0] != 0]; t0 = a[:, 0].min()
            print(f"   window {k} ({paths[-1]}): kw_compact {len(a)} workgroups; phase ends (us, min / mean / max):",
                  [(round(float(((a[:, q] - t0) / 100.0).min()), 1), round(float(((a[:, q] - t0) / 100.0).mean()), 1), round(float(((a[:, q] - t0) / 100.0).max()), 1)) for q in range(6)])
    g.timing_enable(0)
    w = g.timing_samples(10)
    print("   paths:", paths)
    g.close()
    return [round(float(x), 1) for x in w]
print("new edges every window, warm engine :", run(None, False))
if os.environ.get("CHURN_ONLY_FIRST"): sys.exit(0)
print("new edges every window, rebuild only:", run(False, False))
print("same trace every window, warm engine:", run(None, True))
print("same trace every window, rebuild only:", run(False, True))
