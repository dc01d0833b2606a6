"""Multi-GPU ServiceGraph: one process per GPU, the graph hash-sharded by source node.

Sharding
  * every node has one owner shard: hash32(ref) % world (raw-IP outbound nodes: hash of the IP);
  * an event is fed to the owner of its from-endpoint (the source pod; after ReverseDirection the
    resolved destination) — `sg_route` / `route_events` — so every edge accumulator and every CSR
    row lives on exactly one shard and K1 needs no device-side exchange;
  * the join tables are replicated (<= 1.2 MB).

Per window the shards exchange, with torch.distributed (backend "nccl" = RCCL over xGMI):
  1. the raw outbound IPs they saw (tiny all_gather)  -> identical node numbering everywhere;
  2. the integer node statistics (all_gather + local SUM/MAX) -> identical features everywhere;
  3. halo requests: which remote rows each shard needs (destinations of its edges that are owned
     elsewhere and have out-edges);
  4. per SAGE layer: all-to-all of exactly those rows (copied, never reduced => the result is
     bit-identical to the unsharded run).

The driver below is backend-agnostic: `HipBackend` is the product (HIP kernels through the C ABI,
exchange buffers owned by torch so RCCL works on them in place); tests/ plug a CPU stand-in to
exercise the same exchange logic with gloo.
"""
from __future__ import annotations

import time
from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import replay

STAT_SUM_WORDS, STAT_MAX_WORDS = 10, 2


# ------------------------------------------------------------------------------------------------
# ownership / routing (host side; mirrors owner_hash_ref / owner_hash_obip of sg_kernels.h)
# ------------------------------------------------------------------------------------------------
def owner_of_known(node_id: np.ndarray, world: int) -> np.ndarray:
    return replay.hash32(np.asarray(node_id, dtype=np.uint32)) % np.uint32(world)   # ref = KNOWN<<30 | id = id


def owner_of_label(label_idx: np.ndarray, world: int) -> np.ndarray:
    return replay.hash32((np.uint32(1) << np.uint32(30)) | np.asarray(label_idx, dtype=np.uint32)) % np.uint32(world)


def owner_of_obip(ip: np.ndarray, world: int) -> np.ndarray:
    return replay.hash32(np.asarray(ip, dtype=np.uint32) ^ np.uint32(0xA5A5F00D)) % np.uint32(world)


def route_events(ev: np.ndarray, world: int, pod_ip_to_id: dict, svc_ip_to_id: dict) -> np.ndarray:
    """Shard of every event: owner of its from-endpoint (numpy twin of sg_route)."""
    n = len(ev)
    pod_keys = np.fromiter(pod_ip_to_id.keys(), dtype=np.uint32, count=len(pod_ip_to_id))
    pod_vals = np.fromiter(pod_ip_to_id.values(), dtype=np.uint32, count=len(pod_ip_to_id))
    o = np.argsort(pod_keys); pod_keys, pod_vals = pod_keys[o], pod_vals[o]
    svc_keys = np.fromiter(svc_ip_to_id.keys(), dtype=np.uint32, count=len(svc_ip_to_id))
    svc_vals = np.fromiter(svc_ip_to_id.values(), dtype=np.uint32, count=len(svc_ip_to_id))
    o = np.argsort(svc_keys); svc_keys, svc_vals = svc_keys[o], svc_vals[o]

    def lookup(keys, vals, q):
        if len(keys) == 0:
            return np.zeros(len(q), bool), np.zeros(len(q), np.uint32)
        i = np.minimum(np.searchsorted(keys, q), len(keys) - 1)
        hit = keys[i] == q
        return hit, vals[i]
    sp_hit, sp = lookup(pod_keys, pod_vals, ev["saddr"])
    shard = np.where(sp_hit, owner_of_known(sp, world), replay.hash32(ev["saddr"]) % np.uint32(world))
    rev = (ev["flags"] & replay.EV_REVERSE) != 0
    if rev.any():
        ds_hit, ds = lookup(svc_keys, svc_vals, ev["daddr"])
        dp_hit, dp = lookup(pod_keys, pod_vals, ev["daddr"])
        lab = ev["host_label"]
        to_owner = np.where(ds_hit, owner_of_known(ds, world),
                            np.where(dp_hit, owner_of_known(dp, world),
                                     np.where(lab != 0, owner_of_label(np.maximum(lab, 1) - 1, world), owner_of_obip(ev["daddr"], world))))
        shard = np.where(rev & sp_hit, to_owner, shard)
    return shard.astype(np.uint32)


# ------------------------------------------------------------------------------------------------
# communicators
# ------------------------------------------------------------------------------------------------
class DistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def all_to_all_v(self, out: torch.Tensor, inp: torch.Tensor, out_splits: Sequence[int], in_splits: Sequence[int]) -> None:
        _all_to_all_v(out, inp, out_splits, in_splits, self.group)


class ThreadComm:
    """All shards in one process, one thread per shard (validation of G logical shards on one device,
    SURVEY.md §8e): the same collectives through shared memory and a barrier."""

    class Shared:
        def __init__(self, world: int):
            import threading
            self.world = world
            self.slots = [None] * world
            self.barrier = threading.Barrier(world)

    def __init__(self, shared: "ThreadComm.Shared", rank: int):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, obj):
        self.sh.slots[self.rank] = obj
        self.sh.barrier.wait()
        got = list(self.sh.slots)
        self.sh.barrier.wait()
        return got

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        return [x.clone() for x in self._exchange(t)]

    def all_to_all_v(self, out: torch.Tensor, inp: torch.Tensor, out_splits: Sequence[int], in_splits: Sequence[int]) -> None:
        if inp.is_cuda:
            torch.cuda.synchronize(inp.device)
        got = self._exchange((inp, list(in_splits)))
        off = 0
        for r, c in enumerate(out_splits):
            src, splits = got[r]
            so = sum(splits[: self.rank])
            out[off:off + c] = src[so:so + c]; off += c
        if out.is_cuda:
            torch.cuda.synchronize(out.device)
        self.sh.barrier.wait()


def _all_to_all_v(out: torch.Tensor, inp: torch.Tensor, out_splits: Sequence[int], in_splits: Sequence[int], group) -> None:
    """all_to_all_single with uneven splits; falls back to an all_gather emulation where the
    backend has no alltoall (older gloo builds)."""
    try:
        dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=group)
        return
    except (RuntimeError, NotImplementedError):
        pass
    world, me = dist.get_world_size(group), dist.get_rank(group)
    width = inp.shape[1:] if inp.dim() > 1 else ()
    cap = torch.tensor([int(max(in_splits)) if len(in_splits) else 0], dtype=torch.int64, device=inp.device)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
    cap = int(cap.item())
    send = torch.zeros((world, cap) + tuple(width), dtype=inp.dtype, device=inp.device)
    off = 0
    for k, c in enumerate(in_splits):
        send[k, :c] = inp[off:off + c]; off += c
    got = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(got, send, group=group)
    off = 0
    for r, c in enumerate(out_splits):
        out[off:off + c] = got[r][me, :c]; off += c


def run_window(be, comm=None) -> None:
    """One window close on every shard (all shards call it together).  On a GPU backend every torch op
    below runs on the backend's stream, the same one its kernels are enqueued on."""
    import contextlib
    stream = getattr(be, "stream", None)
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        _run_window(be, comm if comm is not None else DistComm())


def _run_window(be, comm) -> None:
    world, me = comm.world, comm.rank
    dev = be.device

    # 1. outbound-IP union -> same OBIP numbering on every shard
    cap = be.max_obip
    mine = be.obip_list()                                    # int64 [n_local] (distinct raw IPs seen here)
    buf = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
    buf[0] = len(mine); buf[1:1 + len(mine)] = mine
    got = comm.all_gather(buf)
    union = torch.cat([g[1:1 + int(g[0].item())] for g in got]) if world > 1 else mine
    be.close(union)

    # 2. node statistics: all_gather, then SUM / MAX locally (integers: exact, order-free)
    flat = be.stats_flat                                     # int64 [ncap*10 | ncap*2], engine writes in place
    st = torch.stack(comm.all_gather(flat))
    ns = be.ncap * STAT_SUM_WORDS
    flat[:ns] = st[:, :ns].sum(dim=0)
    flat[ns:] = st[:, ns:].max(dim=0).values
    be.features()

    # 3. halo requests (ids grouped by owner) -> everyone learns what it must serve
    counts, ids = be.halo_requests()                         # List[int] * world, int64 [sum(counts)]
    call = comm.all_gather(torch.tensor(counts, dtype=torch.int64, device=dev))
    want_from_me = [int(call[r][me].item()) for r in range(world)]          # rows shard r asks of me
    serve_ids = torch.empty(sum(want_from_me), dtype=torch.int64, device=dev)
    comm.all_to_all_v(serve_ids, ids, want_from_me, counts)

    # 4. layers with halo exchange of the produced rows
    for l in range(be.layers):
        be.layer(l)
        rows_out = be.pack(l + 1, serve_ids)                 # float32 [n_serve, 64]
        rows_in = torch.empty((sum(counts), 64), dtype=torch.float32, device=dev)
        comm.all_to_all_v(rows_in, rows_out, counts, want_from_me)
        be.unpack(l + 1, ids, rows_in)
    be.score()


# ------------------------------------------------------------------------------------------------
# product backend: HIP engine + torch-owned exchange buffers
# ------------------------------------------------------------------------------------------------
class HipBackend:
    def __init__(self, g, *, ncap: int, layers: int, world: int, rank: int, device: torch.device, max_obip: int, stream=None):
        self.g, self.ncap, self.layers, self.world, self.rank, self.device, self.max_obip = g, ncap, layers, world, rank, device, max(1, max_obip)
        self.stream = stream if stream is not None else torch.cuda.current_stream(device)
        self.s = self.stream.cuda_stream
        self.stats_flat = torch.zeros(ncap * (STAT_SUM_WORDS + STAT_MAX_WORDS), dtype=torch.int64, device=device)
        self.feat = [torch.zeros((ncap, 64), dtype=torch.float32, device=device) for _ in range(layers)]
        g.bind_buffers(self.stats_flat.data_ptr(), self.stats_flat.data_ptr() + ncap * STAT_SUM_WORDS * 8,
                       [f.data_ptr() for f in self.feat])
        self.ob_list = torch.zeros(self.max_obip, dtype=torch.int32, device=device)
        self.ob_n = torch.zeros(4, dtype=torch.int32, device=device)
        ucap = 1
        while ucap < max(64, self.max_obip * world):
            ucap <<= 1
        self.union = torch.zeros(ucap, dtype=torch.int32, device=device)
        self.union_n = torch.zeros(4, dtype=torch.int32, device=device)
        self.halo_ids = torch.zeros(ncap, dtype=torch.int32, device=device)
        self.halo_counts = torch.zeros(8, dtype=torch.int32, device=device)

    def obip_list(self) -> torch.Tensor:
        self.g.window_obip_list(self.ob_list.data_ptr(), self.max_obip, self.ob_n.data_ptr(), self.s)
        n = int(self.ob_n[0].item())
        return (self.ob_list[:n].to(torch.int64) & 0xFFFFFFFF)

    def close(self, union: torch.Tensor) -> None:
        n = len(union)
        self.union[:n] = union.to(torch.int32) if union.dtype != torch.int32 else union
        self.union_n[0] = n
        self.g.window_close_sharded(self.union.data_ptr(), self.union_n.data_ptr(), self.s)

    def features(self) -> None:
        self.g.window_features(self.s)

    def halo_requests(self):
        self.g.halo_build(self.halo_ids.data_ptr(), self.ncap, self.halo_counts.data_ptr(), self.s)
        counts = [int(x) for x in self.halo_counts[: self.world].tolist()]
        return counts, self.halo_ids[: sum(counts)].to(torch.int64)

    def layer(self, l: int) -> None:
        self.g.window_layer(l, self.s)

    def pack(self, l: int, ids: torch.Tensor) -> torch.Tensor:
        ids32 = ids.to(torch.int32)
        rows = torch.empty((len(ids32), 64), dtype=torch.float32, device=self.device)
        if len(ids32):
            self.g.halo_pack(l, ids32.data_ptr(), len(ids32), rows.data_ptr(), self.s)
        self._keep = ids32
        return rows

    def unpack(self, l: int, ids: torch.Tensor, rows: torch.Tensor) -> None:
        ids32 = ids.to(torch.int32)
        if len(ids32):
            self.g.halo_unpack(l, ids32.data_ptr(), len(ids32), rows.data_ptr(), self.s)
        self._keep2 = (ids32, rows)

    def score(self) -> None:
        self.g.window_score(self.s)


# ------------------------------------------------------------------------------------------------
# weak-scaling bench (called by bench.py under torch.distributed.run)
# ------------------------------------------------------------------------------------------------
def shard_view(topo: replay.Topology, rank: int, world: int) -> replay.Topology:
    """The part of a global topology whose edges this shard owns (source pod owned here)."""
    keep = owner_of_known(topo.edge_src.astype(np.uint32), world) == rank
    return replay.Topology(topo.n_pods, topo.n_svcs, topo.pod_ips, topo.svc_ips, topo.edge_src[keep], topo.edge_dst[keep], topo.seed)


def bench(a, rank: int, world: int, local: int) -> dict:
    from . import engine, weights
    c = replay.CONFIGS[a.config]
    seed = replay.SEED_BASE + a.config
    Ev, L = c["events"], c["layers"]                       # per GPU and window: weak scaling
    P, E = c["pods"] * world, c["edges"] * world
    device = torch.device("cuda", local)
    nb = a.batches or max(2, -(-(320 << 20) // (Ev * 32)))
    topo = replay.make_topology(P, E, seed)
    view = shard_view(topo, rank, world)
    ev_all, labels = replay.make_events(view, Ev * nb, seed + 7919 * (rank + 1), fixed_labels=True)
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=int(len(view.edge_src) * 1.25) + 4096, layers=L,
                            max_labels=max(64, len(labels)), max_outbound_ips=64, device=local, rank=rank, world=world)
    g.set_clock(1_000_000_000, 1_700_000_000_000_000_000)
    g.load_weights(weights.make_weights(L))
    for i in range(topo.n_pods):
        g.upsert_pod(int(topo.pod_ips[i]), i)
    for j in range(topo.n_svcs):
        g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
    g.set_label_count(len(labels))
    ncap = topo.n_nodes + max(64, len(labels)) + 64
    stream = torch.cuda.Stream(device)
    with torch.cuda.stream(stream):
        be = HipBackend(g, ncap=ncap, layers=L, world=world, rank=rank, device=device, max_obip=64, stream=stream)
        dev = [torch.from_numpy(ev_all[i * Ev:(i + 1) * Ev].view(np.uint8).reshape(-1)).to(device) for i in range(nb)]
        torch.cuda.synchronize(device)

        def step(i):
            g.ingest_device(dev[i % nb].data_ptr(), Ev, be.s)
            run_window(be)
            g.window_reset(be.s)

        for i in range(a.warmup):
            step(i)
        g.timing_reset(); g.timing_enable(1 << 1)
        torch.cuda.synchronize(device); dist.barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + i)
        torch.cuda.synchronize(device); dist.barrier()
        dt = time.perf_counter() - t0
        g.timing_enable(0)
        k1_us, k1_n = g.timing(1)
        # edges of one window (untimed)
        g.ingest_device(dev[0].data_ptr(), Ev, be.s)
        run_window(be)
        rows = g.window_read()
        g.window_reset(be.s)
        st = g.stats()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    agg = torch.tensor([float(len(rows)), float(st.events_dropped_cap), float(k1_us)], dtype=torch.float64, device=device)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    Eloc = len(rows)
    alg = 32.0 * Ev + 32.0 * Eloc
    ach = alg / (k1_us * 1e-6) / 1e9 if k1_us > 0 else 0.0
    res = {
        "metric": "L7 edge-events/s ingested->scored service-map", "value": Ev * world * a.steps / dt, "unit": "events/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"C{a.config} x {world}: {P} pods / {topo.n_svcs} services / {E} edges hash-sharded by source pod, "
                               f"{Ev} HTTP l7 events per GPU per window, {L}-layer SAGE + MLP score",
                   "events_per_window": Ev * world, "edges_per_window": int(agg[0].item()), "layers": L,
                   "parallelism": f"{world} shards, RCCL halo all-to-all"},
        "roofline": {"bound": "hbm", "kernel": "k1_resolve_aggregate", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                     "frac": ach / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_us": k1_us,
                     "launches": k1_n, "note": "rank 0's K1"},
    }
    g.close()
    return res
