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

import sys
import time
from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import replay

STAT_SUM_WORDS, STAT_MAX_WORDS = 12, 2


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
    rev = ((ev["flags"] & replay.EV_REVERSE) != 0) & ((ev["flags"] & replay.EV_ALIVE) == 0)   # alive records are never reversed
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
# communicators: three fixed-size collectives are all the window needs (no host round trip)
# ------------------------------------------------------------------------------------------------
class DistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor) -> None:        # out [world, k] <- inp [k]
        try:
            dist.all_gather_into_tensor(out, inp, group=self.group)
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(inp) for _ in range(self.world)]
            dist.all_gather(parts, inp, group=self.group)
            out.copy_(torch.stack(parts))

    def all_reduce_(self, t: torch.Tensor, op: str) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)

    def all_to_all_equal(self, out: torch.Tensor, inp: torch.Tensor) -> None:     # [world, ...] both
        try:
            dist.all_to_all_single(out, inp, group=self.group)
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(inp) for _ in range(self.world)]
            dist.all_gather(parts, inp, group=self.group)
            for r in range(self.world):
                out[r] = parts[r][self.rank]


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

    def _exchange(self, t: torch.Tensor):
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        self.sh.slots[self.rank] = t
        self.sh.barrier.wait()
        got = [x.clone() for x in self.sh.slots]
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        self.sh.barrier.wait()
        return got

    def all_gather_into(self, out, inp):
        out.copy_(torch.stack(self._exchange(inp)))

    def all_reduce_(self, t, op):
        st = torch.stack(self._exchange(t))
        t.copy_(st.sum(dim=0) if op == "sum" else st.max(dim=0).values)

    def all_to_all_equal(self, out, inp):
        got = self._exchange(inp)
        for r in range(self.world):
            out[r] = got[r][self.rank]


def run_window(be, comm=None, fused_reset: bool = False) -> None:
    """One window close on every shard (all shards call it together).  Nothing in here waits for the
    device: every exchange has a fixed size, so the host only enqueues and windows can pipeline.
    On a GPU backend every torch op runs on the backend's stream, the one its kernels are enqueued on."""
    import contextlib
    stream = getattr(be, "stream", None)
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        _run_window(be, comm if comm is not None else DistComm(), fused_reset)


def _run_window(be, comm, fused_reset: bool = False) -> None:
    # 1. raw outbound IPs of every shard -> identical OBIP numbering everywhere
    comm.all_gather_into(be.ob_all, be.ob_local())
    be.close_gathered()
    # 2. integer node statistics: in-place all-reduce (exact, order-free)
    comm.all_reduce_(be.stats_sum, "sum")
    comm.all_reduce_(be.stats_max, "max")
    be.features()
    # 3. halo requests: what I need from each owner <-> what each shard needs from me
    comm.all_to_all_equal(be.serve, be.halo_requests())
    # 4. layers, each followed by the exchange of exactly the requested rows (copied, never reduced)
    for l in range(be.layers):
        be.layer(l)
        comm.all_to_all_equal(be.rows_in, be.pack(l + 1))
        be.unpack(l + 1)
    if fused_reset:
        be.score_reset()          # K5 + window reset in one launch; rows stay in sg_window_rows_buffer()
    else:
        be.score()


# ------------------------------------------------------------------------------------------------
# the same sequence through the C sequencer (alaz_amd/csrc/shard_seq.hpp)
# ------------------------------------------------------------------------------------------------
def run_window_c(be, comm=None, grouped: bool = False, trace=None) -> None:
    """The window sequence of `_run_window`, but issued by the C function the engine library itself runs inside
    sg_window_run_sharded (sg_run_sharded_window, exported by libsgdatastore.so as sgh_run_sharded_window): the stages are
    callbacks into `be`, the collectives callbacks into `comm`.  The HIP engine does not need this detour (it has
    sg_window_run_sharded with RCCL inside); it exists so that the C sequence itself is driven by the CPU tests (gloo).
    grouped: hand the sequencer group_begin / group_end callbacks that behave as ncclGroupStart / ncclGroupEnd do — the all-reduces
    issued between them are only RECORDED and run when the group ends (what the RCCL communicator of sg_window_run_sharded does with
    the SUM and MAX statistics).  trace (a list): the order of the communicator calls the sequencer made."""
    import ctypes as C
    from . import hostlib
    comm = comm if comm is not None else DistComm()
    lib = hostlib.load()
    world, layers = be.world, be.layers
    # fixed exchange buffers (the sequencer passes pointers, not tensors)
    ob_local = be.ob_local().clone()
    bufs = {"ob_local": ob_local, "ob_all": be.ob_all, "serve": be.serve, "rows_in": be.rows_in,
            "req": torch.zeros_like(be.serve), "rows_out": torch.zeros_like(be.rows_in)}
    by_ptr = {}

    def reg(t):
        by_ptr[t.data_ptr()] = t
        return C.c_void_p(t.data_ptr())
    STAGE0 = C.CFUNCTYPE(C.c_int, C.c_void_p)
    STAGE1 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)
    GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)

    class Comm(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("all_gather", GATHER), ("all_reduce_u64", REDUCE), ("all_to_all", GATHER),
                    ("group_begin", STAGE0), ("group_end", STAGE0)]              # (optional grouping of the two statistics all-reduces: null unless `grouped`)

    class Stages(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("layers", C.c_uint32), ("world", C.c_uint32),
                    ("obip_list", STAGE0), ("close_gathered", STAGE0), ("features", STAGE0), ("halo_build", STAGE0),
                    ("layer", STAGE1), ("pack", STAGE1), ("unpack", STAGE1), ("score", STAGE0),
                    ("ob_local", C.c_void_p), ("ob_all", C.c_void_p), ("ob_bytes", C.c_size_t),
                    ("stats_sum", C.c_void_p), ("stats_sum_words", C.c_size_t), ("stats_max", C.c_void_p), ("stats_max_words", C.c_size_t),
                    ("req", C.c_void_p), ("serve", C.c_void_p), ("list_bytes", C.c_size_t),
                    ("rows_out", C.c_void_p), ("rows_in", C.c_void_p), ("rows_bytes", C.c_size_t)]
    errs = []

    def guard(fn):
        def w(*a):
            try:
                fn(*a)
                return 0
            except Exception as ex:                                  # noqa: BLE001 — reported through the return code
                errs.append(ex)
                return -5
        return w

    def per_rank(t):
        return t.numel() * t.element_size() // world

    note = trace.append if trace is not None else (lambda _x: None)
    group = {"open": False, "held": []}

    def c_gather(_ctx, send, recv, nbytes):
        note("all_gather"); comm.all_gather_into(by_ptr[recv], by_ptr[send])
    def c_reduce(_ctx, buf, count, op):
        note("all_reduce_max" if op else "all_reduce_sum")
        if group["open"]: group["held"].append((buf, op))            # (inside a group nothing runs before the group ends)
        else: comm.all_reduce_(by_ptr[buf], "max" if op else "sum")
    def c_a2a(_ctx, send, recv, nbytes):
        note("all_to_all"); comm.all_to_all_equal(by_ptr[recv], by_ptr[send])
    def c_group_begin(_ctx):
        note("group_begin")
        if group["open"]: raise RuntimeError("group_begin inside a group")
        group["open"] = True
    def c_group_end(_ctx):
        note("group_end")
        if not group["open"]: raise RuntimeError("group_end without group_begin")
        group["open"] = False
        held, group["held"] = group["held"], []
        for buf, op in held:
            comm.all_reduce_(by_ptr[buf], "max" if op else "sum")
    cm = Comm(None, GATHER(guard(c_gather)), REDUCE(guard(c_reduce)), GATHER(guard(c_a2a)))
    if grouped:
        cm.group_begin, cm.group_end = STAGE0(guard(c_group_begin)), STAGE0(guard(c_group_end))
    st = Stages()
    st.layers, st.world = layers, world
    st.obip_list = STAGE0(guard(lambda _c: bufs["ob_local"].copy_(be.ob_local())))
    st.close_gathered = STAGE0(guard(lambda _c: be.close_gathered()))
    st.features = STAGE0(guard(lambda _c: be.features()))
    st.halo_build = STAGE0(guard(lambda _c: bufs["req"].copy_(be.halo_requests())))
    st.layer = STAGE1(guard(lambda _c, l: be.layer(l)))
    st.pack = STAGE1(guard(lambda _c, l: bufs["rows_out"].copy_(be.pack(l))))
    st.unpack = STAGE1(guard(lambda _c, l: be.unpack(l)))
    st.score = STAGE0(guard(lambda _c: be.score()))
    st.ob_local, st.ob_all, st.ob_bytes = reg(bufs["ob_local"]), reg(be.ob_all), per_rank(be.ob_all)
    st.stats_sum, st.stats_sum_words = reg(be.stats_sum), be.stats_sum.numel()
    st.stats_max, st.stats_max_words = reg(be.stats_max), be.stats_max.numel()
    st.req, st.serve, st.list_bytes = reg(bufs["req"]), reg(be.serve), per_rank(be.serve)
    st.rows_out, st.rows_in, st.rows_bytes = reg(bufs["rows_out"]), reg(be.rows_in), per_rank(be.rows_in)
    lib.sgh_run_sharded_window.restype = C.c_int
    lib.sgh_run_sharded_window.argtypes = [C.POINTER(Stages), C.POINTER(Comm)]
    rc = lib.sgh_run_sharded_window(C.byref(st), C.byref(cm))
    if errs:
        raise errs[0]
    if rc != 0:
        raise RuntimeError(f"sgh_run_sharded_window returned {rc}")
    if group["open"] or group["held"]:
        raise RuntimeError("the sequencer left a collective group open")


# ------------------------------------------------------------------------------------------------
# product backend: HIP engine + torch-owned exchange buffers
# ------------------------------------------------------------------------------------------------
class HipBackend:
    def __init__(self, g, *, ncap: int, layers: int, world: int, rank: int, device: torch.device, max_obip: int,
                 stream=None, halo_cap: int = 0):
        self.g, self.ncap, self.layers, self.world, self.rank, self.device = g, ncap, layers, world, rank, device
        self.max_obip = max(1, max_obip)
        # rows one shard may request from ONE owner: at most the nodes that owner owns; 2 x the mean share of the node
        # space (hash ownership is even) + slack, never more than all nodes.  A request beyond it is counted in
        # sg_stats.halo_overflow and bench() refuses to report a number then.
        self.capp = max(1, min(ncap, halo_cap if halo_cap > 0 else 2 * -(-ncap // max(world, 1)) + 1024))
        self.stream = stream if stream is not None else torch.cuda.current_stream(device)
        self.s = self.stream.cuda_stream
        z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=device)
        self.stats_flat = z(ncap * (STAT_SUM_WORDS + STAT_MAX_WORDS), dtype=torch.int64)
        self.stats_sum = self.stats_flat[: ncap * STAT_SUM_WORDS]
        self.stats_max = self.stats_flat[ncap * STAT_SUM_WORDS:]
        self.feat = [z(ncap, 64, dtype=torch.float32) for _ in range(layers)]
        g.bind_buffers(self.stats_sum.data_ptr(), self.stats_max.data_ptr(), [f.data_ptr() for f in self.feat])
        self.ob_buf = z(self.max_obip + 1, dtype=torch.int32)                  # [count, ip...]
        self.ob_all = z(world, self.max_obip + 1, dtype=torch.int32)
        self.req = z(world, self.capp + 1, dtype=torch.int32)                  # what I need from each owner
        self.serve = z(world, self.capp + 1, dtype=torch.int32)                # what each shard needs from me
        self.rows_out = z(world, self.capp, 64, dtype=torch.float32)
        self.rows_in = z(world, self.capp, 64, dtype=torch.float32)

    def ob_local(self) -> torch.Tensor:
        self.g.window_obip_list(self.ob_buf.data_ptr() + 4, self.max_obip, self.ob_buf.data_ptr(), self.s)
        return self.ob_buf

    def close_gathered(self) -> None:
        self.g.window_close_gathered(self.ob_all.data_ptr(), self.max_obip + 1, self.world, self.s)

    def features(self) -> None:
        self.g.window_features(self.s)

    def halo_requests(self) -> torch.Tensor:
        self.g.halo_build_padded(self.req.data_ptr(), self.capp, self.s)
        return self.req

    def layer(self, l: int) -> None:
        self.g.window_layer(l, self.s)

    def pack(self, l: int) -> torch.Tensor:
        self.g.halo_pack_padded(l, self.serve.data_ptr(), self.capp, self.rows_out.data_ptr(), self.s)
        return self.rows_out

    def unpack(self, l: int) -> None:
        self.g.halo_unpack_padded(l, self.req.data_ptr(), self.capp, self.rows_in.data_ptr(), self.s)

    def score(self) -> None:
        self.g.window_score(self.s)

    def score_reset(self) -> None:
        self.g.window_score_reset(self.s)


# ------------------------------------------------------------------------------------------------
# weak-scaling bench (called by bench.py under torch.distributed.run)
# ------------------------------------------------------------------------------------------------
def shard_view(topo: replay.Topology, rank: int, world: int) -> replay.Topology:
    """The part of a global topology whose edges this shard owns (source pod owned here)."""
    keep = owner_of_known(topo.edge_src.astype(np.uint32), world) == rank
    return replay.Topology(topo.n_pods, topo.n_svcs, topo.pod_ips, topo.svc_ips, topo.edge_src[keep], topo.edge_dst[keep], topo.seed)


def settle_collectively(step, settle_ms: float, device, sync=lambda: None, trip: int = 8) -> int:
    """Run step(0), step(1), ... in trips of `trip` windows until EVERY rank's clock says settle_ms have passed, and return how many ran:
    the same number on every rank (each window issues collectives, so no rank may run one the others do not join).  The decision to go
    on is itself a collective — the MAX over the ranks' "my clock says continue" flags after each trip; `trip` is even, so windows that
    alternate between two engines / communicators stay aligned across ranks."""
    ts = time.perf_counter(); i = 0
    go = torch.ones(1, dtype=torch.int32, device=device)
    while int(go.item()):
        for _ in range(trip):
            step(i); i += 1
        sync()
        go.fill_(1 if (time.perf_counter() - ts) * 1e3 < settle_ms else 0)
        dist.all_reduce(go, op=dist.ReduceOp.MAX)
    return i


def bench(a, rank: int, world: int, local: int) -> dict:
    """bench.py --gpus N: the line's `value` is the STRONG-scaling figure by default (BASELINE config 4 = config 3's replay over the
    GPUs: the total work is fixed, so value(N) / value(1) is the speed-up the north star asks about); the other mode runs right after
    it, briefly (same steps, no diagnostic pass), and lands in the line's `weak` (or `strong`) object.  SG_BENCH_ONE_MODE=1: only the
    mode --scaling names."""
    import os
    strong = getattr(a, "scaling", "strong") == "strong"
    res = _bench_mode(a, rank, world, local, strong, brief=False)
    if os.environ.get("SG_BENCH_ONE_MODE") != "1" and "error" not in res:
        key = "weak" if strong else "strong"
        try:                                                 # (the second mode must never cost the line its first)
            other = _bench_mode(a, rank, world, local, not strong, brief=True)
            res[key] = {k: other[k] for k in ("value", "unit", "ms_per_step", "steps", "scaling", "comm_us_per_window") if k in other}
            res[key].update({k: other["config"][k] for k in ("events_per_window", "edges_per_window", "dropped_or_misrouted", "halo_overflow", "largest_shard_events")})
            if "error" in other:
                res[key]["error"] = other["error"]
        except Exception as ex:                              # noqa: BLE001
            res[key] = {"error": repr(ex)[:300]}
    return res


def _bench_mode(a, rank: int, world: int, local: int, strong: bool, brief: bool) -> dict:
    """The sharded window on `world` GPUs (bench.py --gpus N under torch.distributed.run).
      strong (--scaling strong, the default)  ONE replay: the same Ev-event windows of the global trace, every event routed to the
                                owner of its source (sg_route's rule) — what "N GPUs on the same job" means
      weak (--scaling weak)     the event volume grows with the GPUs: Ev events per GPU and window, each rank's events drawn from the
                                sources it owns (on a fixed graph the per-GPU graph work shrinks: super-linear in events/s by construction)
      --graph fixed | scaled    the configuration's own graph, or world x pods / edges
      brief                     the timed steps only (no per-group diagnostic pass, no --verify window)
    Two engine instances per GPU alternate windows on two streams, so the exchanges of window w overlap the kernels of window w+1.
    Beside the contract's line: `kernels[]` (per group, from an untimed pass with every group bracketed), `comm_us_per_window` (the
    collectives of one window, by event pairs around each RCCL call), `halo_rows` / `halo_overflow`, and — with --verify — `rows_verified`:
    one window of a global trace through the sharded engines, every rank's rows gathered on rank 0 and compared byte for byte with an
    unsharded engine fed the same events."""
    from . import engine, weights
    import os
    cfgno = 3 if a.config == 4 else a.config               # C4 = C3's graph, sharded
    c = replay.CONFIGS[cfgno]
    seed = replay.SEED_BASE + cfgno
    Ev, L = c["events"], c["layers"]
    gs = world if getattr(a, "graph", "fixed") == "scaled" else 1
    P, E = c["pods"] * gs, c["edges"] * gs
    device = torch.device("cuda", local)
    nb = a.batches or max(2, -(-(320 << 20) // (Ev * 32)))
    topo = replay.make_topology(P, E, seed)
    view = shard_view(topo, rank, world)
    pod_map = {int(ip): i for i, ip in enumerate(topo.pod_ips)}; svc_map = {int(ip): topo.n_pods + j for j, ip in enumerate(topo.svc_ips)}
    if strong:                                             # the SAME global trace on every rank; a rank keeps what is routed to it
        ev_glob, labels = replay.make_events(topo, Ev * nb, seed, fixed_labels=True)
        batches = []
        for i in range(nb):
            b = ev_glob[i * Ev:(i + 1) * Ev]
            batches.append(np.ascontiguousarray(b[route_events(b, world, pod_map, svc_map) == rank]))
        del ev_glob
    else:
        ev_all, labels = replay.make_events(view, Ev * nb, seed + 7919 * (rank + 1), fixed_labels=True)
        batches = [ev_all[i * Ev:(i + 1) * Ev] for i in range(nb)]
    nlab = max(64, len(labels))
    ncap = topo.n_nodes + nlab + 64
    comm = DistComm()
    # default: the window is ONE C call (sg_window_run_sharded: the library issues its collectives on RCCL itself);
    # SG_SHARDED_PY=1 keeps the Python-orchestrated driver (run_window: ~15 ctypes calls + 6 torch.distributed calls per window)
    one_call = os.environ.get("SG_SHARDED_PY") != "1"
    if one_call:                                           # every rank can load librccl, or nobody tries: decided BEFORE ncclCommInitRank
        flag = torch.tensor([1 if engine.RcclComm.probe() else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            one_call = False
            print(f"[rank {rank}] librccl not loadable on some rank; every rank uses the Python driver", file=sys.stderr, flush=True)

    def bcast(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def make_engine(max_edges, r=rank, w=world, max_ev=Ev):
        g = engine.ServiceGraph(max_known_nodes=topo.n_nodes, max_edges=max_edges, layers=L,
                                max_labels=nlab, max_outbound_ips=64, device=local, rank=r, world=w, max_batch=1 << 18,
                                max_window_events=max_ev)
        g.set_clock(1_000_000_000, 1_700_000_000_000_000_000)
        g.load_weights(weights.make_weights(L))
        for i in range(topo.n_pods):
            g.upsert_pod(int(topo.pod_ips[i]), i)
        for j in range(topo.n_svcs):
            g.upsert_service(int(topo.svc_ips[j]), topo.n_pods + j)
        g.set_label_count(len(labels))
        return g
    engs, bes, rcomms, streams = [], [], [], []
    for k in range(2):
        g = make_engine(int(len(view.edge_src) * 1.25) + 4096)
        engs.append(g)
        st = torch.cuda.Stream(device)
        streams.append(st)
        if one_call:
            # one communicator per engine: the two windows in flight do not share a stream.  A rank whose ncclCommInitRank is refused
            # still takes everybody to the Python driver (decided together, so that no rank waits in a collective alone)
            ok = 1
            try:
                rcomms.append(engine.RcclComm(rank, world, local, bcast))
            except engine.ServiceGraphError as ex:
                ok = 0
                print(f"[rank {rank}] sg_comm_create failed ({ex}); falling back to the Python driver", file=sys.stderr, flush=True)
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                for cm in rcomms: cm.close()
                rcomms.clear(); one_call = False
                for g0, st0 in zip(engs, streams):
                    bes.append(HipBackend(g0, ncap=ncap, layers=L, world=world, rank=rank, device=device, max_obip=64, stream=st0))
        else:
            bes.append(HipBackend(g, ncap=ncap, layers=L, world=world, rank=rank, device=device, max_obip=64, stream=st))
    dev = [torch.from_numpy(b.view(np.uint8).reshape(-1)).to(device) for b in batches]
    cnt = [len(b) for b in batches]
    torch.cuda.synchronize(device)
    # A collective that never completes (a rank missing, a link down) would hold the launcher until ITS caller gives up, with nothing to
    # read afterwards: from here on a watchdog dumps every thread's Python stack to stderr and ends the process (torch.distributed.run
    # then ends the other ranks) when the GPU part of the run takes longer than SG_BENCH_WATCHDOG_S seconds (default 600; 0 = off).
    import faulthandler
    wd = float(os.environ.get("SG_BENCH_WATCHDOG_S", "600"))
    if wd > 0:
        faulthandler.dump_traceback_later(wd, exit=True)

    def window(k, d_ptr, n):
        engs[k].ingest_device(d_ptr, n, streams[k].cuda_stream)
        if one_call:
            engs[k].window_run_sharded(rcomms[k], streams[k].cuda_stream)
        else:
            run_window(bes[k], comm, fused_reset=True)

    def step(i):
        window(i & 1, dev[i % nb].data_ptr(), cnt[i % nb])

    if getattr(a, "settle_ms", 0) > 0:                     # untimed real windows until the chip has left its idle power state
        # Every window issues collectives, so every rank must run the SAME number of settle windows: the decision to go on is itself
        # collective (MAX over the ranks' "my clock says continue" flags after each trip of 8 windows — an even count, so the
        # engs[k] / rcomms[k] parity stays aligned).  A per-rank wall-clock test let one rank leave for the barrier while another
        # enqueued eight more windows of all-gather / all-to-all nobody joined.
        settle_collectively(step, a.settle_ms, device, lambda: torch.cuda.synchronize(device))
        dist.barrier()
    for i in range(a.warmup):
        step(i)
    for g in engs:
        g.timing_reset(); g.timing_enable((1 << 1) | (1 << 7))
    torch.cuda.synchronize(device); dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize(device); dist.barrier()
    dt = time.perf_counter() - t0
    for g in engs:
        g.timing_enable(0)
    # per window: all records of a group over the two engines / the timed windows (an engine that keeps warm-window state launches
    # pass B twice per window — the warm attempt and the cold merge, one of which returns at once — and a window's pass B is their sum)
    k1a = sum(g.timing(1)[0] * g.timing(1)[1] for g in engs) / max(1, a.steps); k1b = sum(g.timing(7)[0] * g.timing(7)[1] for g in engs) / max(1, a.steps)
    k1n = sum(g.timing(1)[1] for g in engs)
    # untimed diagnostic pass: every kernel group and every collective bracketed by events (a few us each), per window of rank 0's engines
    nd = 0 if brief else min(10, a.steps)
    for g in engs:
        g.timing_reset(); g.timing_enable(1 if nd else 0)
    for i in range(nd):
        step(i)
    torch.cuda.synchronize(device); dist.barrier()
    for g in engs:
        g.timing_enable(0)
    grp = {}
    for name, kk in (("K1a", 1), ("K1b", 7), ("K2", 2), ("K3-in", 8), ("K3-feat", 3), ("K4", 4), ("K5", 5), ("K6-halo", 6), ("collectives", 9)):
        tot = sum(g.timing(kk)[0] * g.timing(kk)[1] for g in engs)
        grp[name] = tot / nd if nd else 0.0                # us per window (a group may have several records per window)
    # edges / halo of one window (untimed)
    window(0, dev[0].data_ptr(), cnt[0])
    rows = engs[0].window_read()                           # (the counters and the rows survive the fused reset)
    if not one_call:
        pass
    halo = engs[0].halo_counts(world) if one_call else np.zeros(world, dtype=np.uint32)
    st = engs[0].stats()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ev_window = float(np.mean(cnt))                        # this rank's events per window
    agg = torch.tensor([float(len(rows)), float(st.events_dropped_cap + st.events_misrouted), float(st.halo_overflow), float(halo.sum()), ev_window],
                       dtype=torch.float64, device=device)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    gmax = torch.tensor([grp[k] for k in grp] + [float(len(rows)), ev_window], dtype=torch.float64, device=device)
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX)            # the slowest rank's groups (it sets the pace), the largest shard
    dt = float(tmax.item())
    ev_total = float(agg[4].item())                        # events of one window over all ranks (strong: Ev; weak: world x Ev)
    k1_us = float(k1a + k1b)
    alg = 32.0 * ev_window + 32.0 * len(rows)
    ach = alg / (k1_us * 1e-6) / 1e9 if k1_us > 0 else 0.0
    verified = None
    if getattr(a, "verify", False) and not brief:
        verified = _verify_rows(make_engine, window, engs, topo, labels, pod_map, svc_map, rank, world, device, seed, min(Ev, 2_000_000))
    names = list(grp)
    res = {
        "metric": "L7 edge-events/s ingested->scored service-map", "value": ev_total * a.steps / dt, "unit": "events/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"C{a.config}{' x ' + str(world) + ' (graph scaled)' if gs > 1 else ''} device-resident replay: {P} pods / {topo.n_svcs} services / {E} edges "
                               f"hash-sharded by source pod over {world} GPU(s), "
                               + (f"ONE replay of {Ev} HTTP l7 events per window routed over the GPUs (strong scaling)" if strong else f"{Ev} HTTP l7 events per GPU per window (weak scaling in the event volume)")
                               + f", {L}-layer SAGE + MLP score",
                   "events_per_window": int(ev_total), "edges_per_window": int(agg[0].item()), "layers": L,
                   "dropped_or_misrouted": int(agg[1].item()), "halo_overflow": int(agg[2].item()), "halo_rows_per_window": int(agg[3].item()),
                   "largest_shard_edges": int(gmax[len(names)].item()), "largest_shard_events": int(gmax[len(names) + 1].item()),
                   "rccl_ranks": dist.get_world_size(),
                   "parallelism": f"{world} shards, RCCL all-reduce (node stats) + halo all-to-all, 2 windows in flight per GPU",
                   # (for whoever divides this line by the 1-GPU line: that one times ONE window in flight — clean per-kernel durations for its
                   # roofline — and carries the pipelined rate as `overlapped.events_per_s`; this one needs two in flight to overlap its collectives)
                   "value_basis": ("strong scaling: ONE replay of the configuration's events routed over the GPUs (total work fixed); " if strong else
                                   "weak scaling in the event volume (Ev events per GPU and window on the configuration's fixed graph); ")
                                  + "2 windows in flight per GPU; the like-for-like 1-GPU figure is that line's overlapped.events_per_s, not its value",
                   "window_driver": "sg_window_run_sharded (one C call per window, RCCL from the library)" if one_call else "alaz_amd.sharded.run_window (Python, torch.distributed)"},
        "roofline": {"bound": "hbm", "kernel": "K1 resolve_aggregate = " + " + ".join(engs[0].k1_kernels()) + " (rank 0)", "achieved": ach, "peak": 8000.0,
                     "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_us": k1_us,
                     "pass_a_us": float(k1a), "pass_b_us": float(k1b), "launches": int(k1n)},
        # per window, the slowest rank's figure of every group (untimed pass, every group bracketed by events; with two windows in flight the
        # groups of the two engines overlap, so they do not add up to ms_per_step)
        "kernels": [{"name": n_, "us_per_window_max_rank": round(float(gmax[i].item()), 2)} for i, n_ in enumerate(names) if n_ != "collectives"],
        "comm_us_per_window": None if brief else round(float(gmax[names.index("collectives")].item()), 2),
        "rows_verified": verified,
    }
    if res["config"]["halo_overflow"]:
        res["error"] = "halo_overflow != 0: rows are incomplete, the number above is not a result"
    elif res["config"]["dropped_or_misrouted"]:
        res["error"] = "dropped_or_misrouted != 0: events were lost to a capacity or routed to the wrong shard, the number above is not a result"
    for cm in rcomms:
        cm.close()
    for g in engs:
        g.close()
    faulthandler.cancel_dump_traceback_later()
    return res


def _verify_rows(make_engine, window, engs, topo, labels, pod_map, svc_map, rank, world, device, seed, nver) -> bool:
    """One window of a GLOBAL trace through the sharded engines (every rank feeds what is routed to it), all rows gathered on rank 0 and
    compared byte for byte with an unsharded engine fed the whole trace: routing, collectives and rows in one check (untimed)."""
    ev, _ = replay.make_events(topo, nver, seed + 4242, fixed_labels=True)
    mine = np.ascontiguousarray(ev[route_events(ev, world, pod_map, svc_map) == rank])
    d = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy()).to(device)
    window(0, d.data_ptr(), len(mine))
    rows = engs[0].window_read().copy()
    box = [None] * world if rank == 0 else None
    dist.gather_object(rows.tobytes(), box, dst=0)
    ok = True
    if rank == 0:
        from . import engine
        got = np.concatenate([np.frombuffer(b, dtype=engine.EDGE_OUT_DTYPE) for b in box])
        ref = make_engine(int(len(topo.edge_src) * 1.25) + 4096, r=0, w=1, max_ev=nver)
        dall = torch.from_numpy(ev.view(np.uint8).reshape(-1).copy()).to(device)
        ref.ingest_device(dall.data_ptr(), len(ev), 0); ref.window_run(0)
        torch.cuda.synchronize(device)
        want = ref.window_read().copy()
        ref.close()
        key = lambda x: np.lexsort((x["to_ref"], x["from_ref"]))
        ok = len(got) == len(want) and got[key(got)].tobytes() == want[key(want)].tobytes()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.broadcast(flag, src=0)
    return bool(flag.item())
