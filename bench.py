#!/usr/bin/env python3
"""bench.py — L7 edge-events/s ingested -> scored service map on MI355X (BASELINE.json metric).

A step = one window of the hot path over one batch of synthetic events that is already resident
in HBM: K1 resolve_aggregate over the batch, then K2..K5 (CSR build, node/edge features, SAGE
layer(s), edge scores) and the window reset.  N=1 workload = BASELINE config 2 (1k pods / 500
services / 50k edges / 1M events per window, L=1).  Steps cycle through a ring of distinct batches
larger than the 256 MiB Infinity Cache, so every step streams its events from HBM.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel K1 (algorithmic bytes
32*Ev + 32*E per launch, SURVEY.md §8d / DESIGN.md), timed with HIP events on the launch stream.
`cpu_baseline` is the CPU oracle (a C restatement of the reference's aggregator path; the Go
binary cannot be built here) timed single-threaded on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3])
    ap.add_argument("--batches", type=int, default=0, help="distinct event batches in the HBM ring (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--windows", type=int, default=1, help="window slots in flight for the timed region (sg_config.windows_in_flight); "
                                                            "1 keeps the per-kernel timings uncontended")
    ap.add_argument("--overlap-windows", type=int, default=4, help="extra diagnostic pass with this many windows in flight (0 = skip)")
    ap.add_argument("--graph", choices=["fixed", "scaled"], default="fixed",
                    help="N > 1: 'fixed' shards the configuration's own graph over the N GPUs (what BASELINE's multi-GPU configurations "
                         "do with theirs) and scales the event volume, 1 M events per GPU per window; 'scaled' also grows the graph N-fold")
    ap.add_argument("--profile-mode", action="store_true", help="only warm-up + the timed steps (no diagnostic passes, no CPU baseline): "
                                                                  "the run rocprofv3 wraps, so its per-kernel averages are those of the timed region")
    return ap.parse_args()


def cpu_baseline(topo, events, labels, layers, seconds):
    """The reference's CPU path restated (oracle/sg_oracle.c): full 1096-byte records through
    processL7 -> processHttpEvent -> setFromToV2 -> PersistRequest, then the window close.
    `value` is one core; `all_cores` runs one independent oracle per host core on the same sample
    (no shared tables, so it is an upper bound for a lock-sharing aggregator, data.go:812-825)."""
    import threading
    from alaz_amd import replay, weights
    from oracle import pyoracle
    sample = events[: min(len(events), 200_000)]
    wire = replay.to_wire(sample, labels)
    W = weights.make_weights(layers)
    ops = topo.k8s_ops()

    def run(secs, out, k):
        o = pyoracle.Oracle(1_000_000_000, 1_700_000_000_000_000_000)
        o.apply_ops(ops)
        done, t0 = 0, time.perf_counter()
        while True:
            o.l7_wire(wire)             # ctypes releases the GIL for the whole call
            o.window_close(W, layers)
            done += len(sample)
            if time.perf_counter() - t0 >= secs:
                break
        out[k] = done
        o.close()

    one = [0]
    t0 = time.perf_counter(); run(seconds, one, 0); dt = time.perf_counter() - t0
    res = {"value": one[0] / dt, "unit": "events/s", "cores": 1, "kind": "port",
           "sample": f"{len(sample)} events of the same workload as full 1096-B l7_event records, repeated "
                     f"{one[0] // len(sample)}x ({dt:.1f} s): C restatement of processL7..PersistRequest + window close "
                     "(oracle/sg_oracle.c); the Go aggregator itself cannot be built here (no Go toolchain)"}
    nc = max(1, min(os.cpu_count() or 1, 64))
    if nc > 1:
        outs = [0] * nc
        th = [threading.Thread(target=run, args=(max(2.0, seconds / 2), outs, k)) for k in range(nc)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dtm = time.perf_counter() - t0
        res["all_cores"] = {"value": sum(outs) / dtm, "unit": "events/s", "cores": nc, "host_cpus": os.cpu_count(),
                            "sample": f"{nc} independent oracle instances (one thread each), same sample, {dtm:.1f} s"}
    return res


def pmc_traffic(config):
    """HBM bytes per K1 launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs of `bench.py --profile-mode`, tools/gpu_pmc.sh; gfx950 corrections per
    MI355X_MICROARCH.md are applied by tools/pmc_summary.py).  Counters cannot be read from inside the
    process being timed, so this is the last measured value for this workload, or null."""
    path = os.path.join(ROOT, "profiles", f"pmc_k1_c{config}.json")
    try:
        with open(path) as f:
            j = json.load(f)
        return float(j["k1_total_hbm_bytes"]), f"profiles/pmc_k1_c{config}.json ({j.get('round', '?')})"
    except Exception:
        return None, None


def measured_copy_gbs(torch):
    """Device-to-device copy rate of this box (read + write bytes / time), SURVEY.md 8(d)."""
    n = 1 << 30
    x = torch.empty(n, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
    y.copy_(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    return 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from alaz_amd import engine, replay, weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local)
    force_sharded = os.environ.get("SG_FORCE_SHARDED") == "1"      # exercise the multi-GPU code path at world = 1
    # the contract is ONE line on stdout: libraries that print there (RCCL writes its version banner to stdout when the
    # first communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1 or force_sharded:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        if world > 1 or force_sharded:
            from alaz_amd import sharded
            res = sharded.bench(a, rank, world, local)
        else:
            res = bench_single(a, local)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1 or force_sharded:
        dist.barrier()
        dist.destroy_process_group()


def bench_single(a, device):
    import torch
    from alaz_amd import engine, replay, weights

    c = replay.CONFIGS[a.config]
    seed = replay.SEED_BASE + a.config
    Ev, L = c["events"], c["layers"]
    nb = a.batches or max(2, -(-(320 << 20) // (Ev * 32)))          # ring >= 320 MB > 256 MiB Infinity Cache
    topo = replay.make_topology(c["pods"], c["edges"], seed)
    ev_all, labels = replay.make_events(topo, Ev * nb, seed)
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(c["edges"] * 1.25) + 4096, layers=L,
                            max_labels=max(64, len(labels)), max_outbound_ips=64, device=device, max_batch=1 << 18,
                            max_window_events=Ev, windows_in_flight=a.windows)
    g.set_clock(1_000_000_000, 1_700_000_000_000_000_000)
    g.load_weights(weights.make_weights(L))
    for i in range(topo.n_pods):
        g.upsert_pod(int(topo.pod_ips[i]), i)
    for j in range(topo.n_svcs):
        g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
    g.set_label_count(len(labels))

    s = 0          # NULL stream argument: the engine enqueues every window on its own slot's stream
    dev = [torch.from_numpy(ev_all[i * Ev:(i + 1) * Ev].view(np.uint8).reshape(-1)).cuda() for i in range(nb)]
    torch.cuda.synchronize()

    def step(i):
        g.ingest_device(dev[i % nb].data_ptr(), Ev, s)
        g.window_run(s)

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    g.timing_reset(); g.timing_enable((1 << 1) | (1 << 7))  # HIP events around every K1 launch (pass A + pass B), on their stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g.timing_enable(0)
    k1a, k1b = g.timing(1), g.timing(7)
    # untimed diagnostic pass: per-group durations of the rest of the window pipeline
    k_us = {}
    if not a.profile_mode:
        g.timing_reset(); g.timing_enable(1)
        for i in range(min(20, a.steps)):
            step(i)
        torch.cuda.synchronize()
        g.timing_enable(0)
        k_us = {k: g.timing(k) for k in range(1, 6)}

    # one untimed window with copy-out: how many edges / nodes a window of this workload has
    g.ingest_device(dev[0].data_ptr(), Ev, s)
    torch.cuda.synchronize()
    rows = g.flush_window()
    st = g.stats()
    E = int(st.last_window_edges)
    k1_us = k1a[0] + k1b[0]                                  # K1 = k1a_partition (per batch) + k1b_merge (per window)
    alg_bytes = 32.0 * Ev + 32.0 * E
    achieved = alg_bytes / (k1_us * 1e-6) / 1e9 if k1_us > 0 else 0.0
    traffic, traffic_src = pmc_traffic(a.config)
    copy_gbs = None if a.profile_mode else measured_copy_gbs(torch)
    res = {
        "metric": "L7 edge-events/s ingested->scored service-map", "value": Ev * a.steps / dt, "unit": "events/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"C{a.config}: {c['pods']} pods / {topo.n_svcs} services / {c['edges']} edges, "
                               f"{Ev} HTTP l7 events per window, {L}-layer SAGE + MLP score; {nb}-batch HBM ring",
                   "events_per_window": Ev, "edges_per_window": E, "nodes": int(st.last_window_nodes), "layers": L,
                   "windows_in_flight": a.windows, "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "kernel": "K1 resolve_aggregate = k1a_partition + k1b_merge", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src, "measured_copy_GBs": copy_gbs,
                     "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": k1_us,
                     "k1a_partition_us": k1a[0], "k1b_merge_us": k1b[0], "launches": k1a[1]},
        "kernel_group_us": {"K1a": round(k1a[0], 2), "K1b": round(k1b[0], 2), **{f"K{k}": round(v[0], 2) for k, v in k_us.items() if k > 1}},
    }
    # diagnostic (never `value`): the same steps with several windows in flight inside one engine — the
    # latency-bound close of window w overlaps the ingest of window w+1 (per-kernel durations stretch,
    # throughput rises); no timing events in this pass
    if a.overlap_windows > 1 and not a.profile_mode:
        g2 = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(c["edges"] * 1.25) + 4096, layers=L,
                                 max_labels=max(64, len(labels)), max_outbound_ips=64, device=device, max_batch=1 << 18,
                                 max_window_events=Ev, windows_in_flight=a.overlap_windows)
        g2.set_clock(1_000_000_000, 1_700_000_000_000_000_000)
        g2.load_weights(weights.make_weights(L))
        for i in range(topo.n_pods):
            g2.upsert_pod(int(topo.pod_ips[i]), i)
        for j in range(topo.n_svcs):
            g2.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
        g2.set_label_count(len(labels))
        for i in range(a.warmup):
            g2.ingest_device(dev[i % nb].data_ptr(), Ev, 0); g2.window_run(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.steps):
            g2.ingest_device(dev[(a.warmup + i) % nb].data_ptr(), Ev, 0); g2.window_run(0)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        res["overlapped"] = {"windows_in_flight": a.overlap_windows, "events_per_s": Ev * a.steps / dt2, "ms_per_step": dt2 / a.steps * 1e3}
        g2.close()
    # diagnostic (never `value`): the same windows fed from host memory through sg_ingest
    # (pinned staging ring + H2D over PCIe), DESIGN.md "PCIe-inclusive rate"
    hs = 0 if a.profile_mode else min(10, a.steps)
    chunk = 1 << 18
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(hs):
        b = ev_all[(i % nb) * Ev:((i % nb) + 1) * Ev]
        for j in range(0, Ev, chunk):
            while g.ingest(b[j:j + chunk]) != 0:
                pass
        g.window_run()
    torch.cuda.synchronize()
    if hs:
        res["host_fed_events_per_s"] = Ev * hs / (time.perf_counter() - t0)
    if not a.no_cpu_baseline and not a.profile_mode:
        res["cpu_baseline"] = cpu_baseline(topo, ev_all[:Ev], labels, L, a.cpu_seconds)
    g.close()
    return res


if __name__ == "__main__":
    main()
