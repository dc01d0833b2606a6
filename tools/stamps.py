#!/usr/bin/env python3
"""Phase timeline of the K1 kernels from the SG_STAMP buffer (run with SG_ABLATE=0x100).
Prints, per kernel, for every stamp: min / mean / max over workgroups, in us since the earliest
workgroup start of that launch."""
import os, sys
os.environ["SG_ABLATE"] = os.environ.get("SG_ABLATE", "0x100")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from alaz_amd import engine, replay, weights

cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 2
c = replay.CONFIGS[cfgno]; seed = replay.SEED_BASE + cfgno
topo = replay.make_topology(c["pods"], c["edges"], seed)
Ev = c["events"]
ev, labels = replay.make_events(topo, Ev * 3, seed)
g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(c["edges"] * 1.25) + 4096, layers=c["layers"],
                        max_labels=max(64, len(labels)), max_outbound_ips=64, max_batch=1 << 18, max_window_events=Ev)
g.set_clock(1_000_000_000, 1_700_000_000_000_000_000); g.load_weights(weights.make_weights(c["layers"]))
for i in range(topo.n_pods): g.upsert_pod(int(topo.pod_ips[i]), i)
for j in range(topo.n_svcs): g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
g.set_label_count(len(labels))
dev = [torch.from_numpy(ev[i * Ev:(i + 1) * Ev].view(np.uint8).reshape(-1)).cuda() for i in range(3)]
for i in range(6):
    g.ingest_device(dev[i % 3].data_ptr(), Ev, 0); g.window_run(0)
torch.cuda.synchronize()
st = g.debug_stamps()
print("geometry", g.geometry())
names = {0: ["start", "tables staged+barrier", "-", "events folded", "barrier", "cache flushed, headers", "stats"],
         1: ["start", "headers+table zeroed", "-", "merged+barrier", "overflow list", "compacted+barrier"]}
for kid, kname in zip((0, 1), g.k1_kernels()):
    a = st[kid].astype(np.int64)
    live = a[:, 0] != 0
    a = a[live]
    if not len(a):
        print(kname, "no stamps"); continue
    t0 = a[:, 0].min()
    print(f"{kname}: {len(a)} workgroups; 100 MHz ticks -> us")
    for k, nm in enumerate(names[kid]):
        if nm == "-": continue
        col = (a[:, k] - t0) / 100.0
        print(f"  {k} {nm:<22} min {col.min():7.2f}  mean {col.mean():7.2f}  max {col.max():7.2f}")
    st_, en_ = (a[:, 0] - t0) / 100.0, (a[:, len(names[kid]) - 1] - t0) / 100.0
    print("  start percentiles 10/25/50/75/90:", np.round(np.percentile(st_, [10, 25, 50, 75, 90]), 1), " duration percentiles:", np.round(np.percentile(en_ - st_, [10, 50, 90]), 1))
    if kid == 1 and a[:, 7].max() > 0:                              # where and when the workgroups of pass B start: (XCC, SE, SH, CU) of every workgroup
        hw = a[:, 7]; cu = ((hw >> 32) & 15) * 1024 + ((hw >> 13) & 7) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)
        idx = np.flatnonzero(live)
        print("  distinct CUs:", len(np.unique(cu)), " start by block-index quarter (mean us):", [round(float(st_[(idx >= q * len(idx) // 4) & (idx < (q + 1) * len(idx) // 4)].mean()), 1) for q in range(4)])
        order = np.argsort(st_); first = {}
        for i in order: first.setdefault(int(cu[i]), []).append(round(float(st_[i]), 1))
        ks = sorted(first)[:6]
        print("  starts on six CUs:", {k: first[k] for k in ks})
        nth = np.array([[v[j] if len(v) > j else np.nan for j in range(4)] for v in first.values()])
        print("  n-th workgroup on its CU starts at (mean us):", np.round(np.nanmean(nth, axis=0), 1), " workgroups per CU min/max:", min(len(v) for v in first.values()), max(len(v) for v in first.values()))
    if kid == 1 and a[:, 7].max() > 0:
        # per CU: when its last workgroup ends, how long its slots sit empty; by order of arrival on the CU: duration; the longest workgroups
        dur = en_ - st_; cus = {}
        for i in np.argsort(st_): cus.setdefault(int(cu[i]), []).append(i)
        last_end = np.array([max(en_[i] for i in v) for v in cus.values()]); busy = np.array([sum(dur[i] for i in v) for v in cus.values()])
        print("  per CU: last end percentiles 0/10/50/90/100:", np.round(np.percentile(last_end, [0, 10, 50, 90, 100]), 1), " sum of workgroup durations 10/50/90:", np.round(np.percentile(busy, [10, 50, 90]), 1))
        byord = [[dur[v[j]] for v in cus.values() if len(v) > j] for j in range(4)]
        print("  duration of the n-th workgroup on its CU (mean us):", [round(float(np.mean(x)), 1) if x else None for x in byord])
        rec = a[:, 6].astype(np.float64)
        if rec.max() > 0:
            print("  narrow records per workgroup min/mean/max:", int(rec.min()), round(float(rec.mean())), int(rec.max()), " corr(duration, records):", round(float(np.corrcoef(dur, rec)[0, 1]), 3))
            ph = (a[:, 3] - a[:, 1]) / 100.0
            print("  merge phase us vs records, by record-count quintile:", [(int(rec[q].mean()), round(float(ph[q].mean()), 1)) for q in np.array_split(np.argsort(rec), 5)])
        top = np.argsort(-dur)[:8]
        print("  longest workgroups (block, start, phases header/merge/wide/out, records):", [(int(np.flatnonzero(live)[i]), round(float(st_[i]), 1), [round(float((a[i, k2] - a[i, k1]) / 100.0), 1) for k1, k2 in ((0, 1), (1, 3), (3, 4), (4, 5))], int(rec[i])) for i in top])
    ts = np.arange(0, en_.max(), 10.0)
    print("  workgroups running at t =", {int(t): int(((st_ <= t) & (en_ > t)).sum()) for t in ts})
if st[2][:, 0].max() > 0:                                            # kw_compact (warm windows): start, loads in + ballots, look-back done, written + folded, row pointers + barrier, flushed
    a = st[2].astype(np.int64); a = a[a[:, 0] != 0]; t0 = a[:, 0].min()
    print(f"kw_compact: {len(a)} workgroups")
    for k, nm in enumerate(("start", "loads in, ballots", "look-back done", "written + folded", "row pointers, barrier", "flushed")):
        col = (a[:, k] - t0) / 100.0
        print(f"  {k} {nm:<22} min {col.min():7.2f}  mean {col.mean():7.2f}  max {col.max():7.2f}")
    dur = (a[:, 5] - a[:, 0]) / 100.0
    print("  duration percentiles 10/50/90:", np.round(np.percentile(dur, [10, 50, 90]), 1), " phase means:", [round(float(((a[:, k + 1] - a[:, k]) / 100.0).mean()), 2) for k in range(5)])
if st[3][:, 0].max() > 0:                                            # k3_in_part: start, LDS zeroed + barrier, slice scanned (thread 0), barrier passed, partials written
    a = st[3].astype(np.int64); a = a[a[:, 0] != 0]; t0 = a[:, 0].min()
    print(f"k3_in_part: {len(a)} workgroups")
    for k, nm in enumerate(("start", "zeroed+barrier", "scanned (thread 0)", "barrier", "written")):
        col = (a[:, k] - t0) / 100.0
        print(f"  {k} {nm:<22} min {col.min():7.2f}  mean {col.mean():7.2f}  max {col.max():7.2f}")
if st[2][:, 0].max() > 0:                                            # narrow pass A: wave 0's accumulated clock ticks per phase (kernel slot 2)
    a = st[2].astype(np.int64); a = a[a[:, 0] != 0]
    for k, nm in enumerate(("P1 fold", "barrier-1 wait", "P2 scan", "P3 drop", "barrier-3 wait", "P4 copy-out", "(P1: load waits)", "(P1: group-0 join)")):
        col = a[:, k] / 100.0
        print(f"  sum {nm:<16} min {col.min():7.2f}  mean {col.mean():7.2f}  max {col.max():7.2f}")
g.close()
