"""CPU tests of the multi-GPU path: 2 processes, gloo, the real exchange logic of
alaz_amd.sharded.run_window driven by a numpy stand-in backend (tests/np_backend.py), checked
against the unsharded oracle.  Also the host-side routing rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alaz_amd import replay, sharded, weights
from oracle import pyoracle

CLOCK = (1_000_000_000, 1_700_000_000_000_000_000)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _trace(layers):
    topo = replay.make_topology(40, 300, seed=201)
    ev, labels = replay.make_events(topo, 6000, seed=202, mixed=True, with_raw_outbound=True, with_reverse=True, fixed_labels=True)
    # f-2: alive-connection records ride the same shards (some with a stray REVERSE flag, which must be ignored)
    rng = np.random.default_rng(203)
    al = np.zeros(400, dtype=replay.EVENT_DTYPE)
    al["flags"] = replay.EV_ALIVE; al["flags"][::3] |= replay.EV_REVERSE
    al["saddr"] = topo.pod_ips[rng.integers(0, topo.n_pods, len(al))]
    al["daddr"] = np.where(rng.random(len(al)) < 0.7, topo.svc_ips[rng.integers(0, topo.n_svcs, len(al))], 0x5DB8D800 + rng.integers(0, 10, len(al))).astype(np.uint32)
    ev = np.concatenate([ev[:3000], al, ev[3000:]])
    return topo, ev, labels


def _worker(rank, world, port, layers, q, c_sequencer=False):
    from tests.np_backend import NumpyBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        topo, ev, labels = _trace(layers)
        pod = {int(ip): i for i, ip in enumerate(topo.pod_ips)}
        svc = {int(ip): topo.n_pods + j for j, ip in enumerate(topo.svc_ips)}
        kind = [1] * topo.n_pods + [2] * topo.n_svcs
        shard = sharded.route_events(ev, world, pod, svc)
        be = NumpyBackend(pod_ip_to_id=pod, svc_ip_to_id=svc, kind=kind, n_labels=len(labels), weights=weights.make_weights(layers),
                          layers=layers, rank=rank, world=world, ncap=topo.n_nodes + len(labels) + 64)
        be.ingest(ev[shard == rank])
        calls = []
        if c_sequencer:                                      # the C function sg_window_run_sharded runs (shard_seq.hpp), collectives = gloo
            sharded.run_window_c(be, grouped=c_sequencer == "grouped", trace=calls)
        else: sharded.run_window(be)
        q.put((rank, be.rows, be.misrouted, be.N, [int(x) for x in be.ob], [int(x) for x in be.alive_csr], calls))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,layers,c_sequencer", [(2, 1, False), (2, 1, True), (2, 2, False), (2, 2, True),
                                                      (3, 2, True), (4, 2, True), (4, 1, False), (2, 2, "grouped"), (3, 1, "grouped")])
def test_shards_equal_unsharded_oracle(world, layers, c_sequencer):
    """Two, three and four gloo processes close a window together — through the Python driver (run_window) and through the C
    sequencer the engine's one-call entry point sg_window_run_sharded is built on (sgh_run_sharded_window) — and must reproduce
    the unsharded oracle row for row.  (Three: ownership modulo a number that is not a power of two; four: every shard serves
    halo rows to several peers in one all-to-all.  "grouped": the communicator has group_begin / group_end with ncclGroupStart / End
    semantics — all-reduces issued inside a group run only when it ends — as the RCCL communicator of sg_window_run_sharded has:
    the sequencer must not read the statistics before the group has ended, and must bracket exactly the SUM and the MAX.)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, layers, q, c_sequencer)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if c_sequencer:                                          # the communicator calls of one window, in the order shard_seq.hpp documents
        grp = c_sequencer == "grouped"
        seq = ["all_gather"] + (["group_begin"] if grp else []) + ["all_reduce_sum", "all_reduce_max"] + (["group_end"] if grp else []) \
            + ["all_to_all"] * (1 + layers)
        assert all(g[6] == seq for g in got)
    topo, ev, labels = _trace(layers)
    o = pyoracle.Oracle(*CLOCK); o.apply_ops(topo.k8s_ops()); o.packed(ev, labels); o.window_close(weights.make_weights(layers), layers)
    want = o.edge_rows()
    NK, NL = o.n_known, len(labels)

    def dense(ref):
        t, v = ref >> 30, ref & 0x3FFFFFFF
        return int(v if t == 0 else (NK + v if t == 1 else NK + NL + v))
    ref_rows = {(dense(int(r["from_ref"])), dense(int(r["to_ref"]))): r for r in want}
    rows = [r for g in got for r in g[1]]
    assert sum(g[2] for g in got) == 0                       # nothing misrouted by route_events
    assert all(g[3] == o.n_nodes for g in got)               # same node numbering on every shard
    assert all(g[4] == [int(x) for x in o.outbound_ips()] for g in got)
    assert len(rows) == len(ref_rows) and len({(r[0], r[1]) for r in rows}) == len(rows)   # every edge on exactly one shard
    for f, t, acc, score, z, er in rows:
        w = ref_rows[(f, t)]
        assert acc == (int(w["count"]), int(w["err_count"]), int(w["sum_ns"]), int(w["max_ns"]), int(w["sumsq_us"]))
        assert abs(score - float(w["score"])) <= 1e-5 and abs(z - float(w["lat_z"])) <= 1e-5 * max(1.0, abs(float(w["lat_z"])))
        assert er == float(w["err_ratio"])
    # both shards actually own edges, and some edges need halo rows
    assert all(len(g[1]) > 0 for g in got)
    # alive counts: per edge as in the oracle (rows and alive_csr are both in the shard's CSR order)
    for g in got:
        for (f, t, *_), a in zip(g[1], g[5]):
            assert a == int(ref_rows[(f, t)]["alive"])
    assert sum(sum(g[5]) for g in got) == int(want["alive"].sum()) == 400


def test_route_events_matches_the_ownership_rule():
    topo, ev, labels = _trace(1)
    pod = {int(ip): i for i, ip in enumerate(topo.pod_ips)}
    svc = {int(ip): topo.n_pods + j for j, ip in enumerate(topo.svc_ips)}
    for world in (2, 4, 8):
        sh = sharded.route_events(ev, world, pod, svc)
        plain = ((ev["flags"] & replay.EV_REVERSE) == 0) | ((ev["flags"] & replay.EV_ALIVE) != 0)
        known = np.isin(ev["saddr"], topo.pod_ips)
        ids = (ev["saddr"][plain & known] - replay.POD_IP_BASE).astype(np.uint32)
        assert np.array_equal(sh[plain & known], replay.hash32(ids) % np.uint32(world))
        assert sh.max() < world
        # balance: no shard is starved
        assert np.bincount(sh, minlength=world).min() > len(ev) / world / 3


def test_collectives_on_a_single_process_group():
    """The three collectives of the window on a 1-process group: identity."""
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        c = sharded.DistComm()
        x = torch.arange(12, dtype=torch.float32).reshape(1, 3, 4); y = torch.empty_like(x)
        c.all_to_all_equal(y, x)
        assert torch.equal(x, y)
        g = torch.empty(1, 5, dtype=torch.int64); c.all_gather_into(g, torch.arange(5)); assert g[0].tolist() == [0, 1, 2, 3, 4]
        t = torch.tensor([3, 4]); c.all_reduce_(t, "max"); assert t.tolist() == [3, 4]
    finally:
        dist.destroy_process_group()


def test_thread_comm_runs_the_same_pipeline_without_processes():
    """ThreadComm (all shards in one process) gives the same result as the 2-process gloo run."""
    import threading
    from tests.np_backend import NumpyBackend
    world, layers = 2, 1
    topo, ev, labels = _trace(layers)
    pod = {int(ip): i for i, ip in enumerate(topo.pod_ips)}
    svc = {int(ip): topo.n_pods + j for j, ip in enumerate(topo.svc_ips)}
    shard = sharded.route_events(ev, world, pod, svc)
    shared = sharded.ThreadComm.Shared(world)
    bes = [NumpyBackend(pod_ip_to_id=pod, svc_ip_to_id=svc, kind=[1] * topo.n_pods + [2] * topo.n_svcs, n_labels=len(labels),
                        weights=weights.make_weights(layers), layers=layers, rank=r, world=world, ncap=topo.n_nodes + len(labels) + 64) for r in range(world)]
    for r in range(world):
        bes[r].ingest(ev[shard == r])
    ths = [threading.Thread(target=sharded.run_window, args=(bes[r], sharded.ThreadComm(shared, r))) for r in range(world)]
    for t in ths: t.start()
    for t in ths: t.join(timeout=120)
    o = pyoracle.Oracle(*CLOCK); o.apply_ops(topo.k8s_ops()); o.packed(ev, labels); o.window_close(weights.make_weights(layers), layers)
    rows = [r for b in bes for r in b.rows]
    assert len(rows) == len(o.edge_rows())
    assert sorted(r[2] for r in rows) == sorted((int(w["count"]), int(w["err_count"]), int(w["sum_ns"]), int(w["max_ns"]), int(w["sumsq_us"])) for w in o.edge_rows())


def test_halo_capacity_holds_for_the_bench_graph_at_2_4_8_shards():
    """The sharded bench refuses to report a number when a shard asks one owner for more rows than the fixed-size exchange holds
    (halo_overflow).  The capacity both drivers derive — 2 x ceil(ncap / world) + 1024 rows per (shard, owner) pair — against what
    BASELINE config 3's graph (= config 4 sharded) really asks for: the distinct destinations of a shard's edges that another shard owns
    AND that have out-edges themselves (only those rows are needed: a node without out-edges is nobody's neighbour source)."""
    c = replay.CONFIGS[3]
    topo = replay.make_topology(c["pods"], c["edges"], replay.SEED_BASE + 3)
    src, dst = topo.edge_src.astype(np.uint32), topo.edge_dst.astype(np.uint32)
    has_out = np.zeros(topo.n_nodes, dtype=bool); has_out[src] = True
    ncap = topo.n_nodes + 64 + 64                                          # as sharded.bench sizes it
    for world in (2, 4, 8):
        capp = max(1, min(ncap, 2 * -(-ncap // world) + 1024))
        so, do = sharded.owner_of_known(src, world), sharded.owner_of_known(dst, world)
        need = (so != do) & has_out[dst]
        worst = 0
        for s in range(world):
            m = need & (so == s)
            pair = np.unique(dst[m].astype(np.uint64) * world + do[m])       # distinct (destination, its owner) of shard s
            worst = max(worst, int(np.bincount((pair % world).astype(np.int64), minlength=world).max()))
        own = np.bincount(sharded.owner_of_known(np.arange(topo.n_nodes, dtype=np.uint32), world), minlength=world)
        assert worst <= own.max() <= capp, (world, worst, int(own.max()), capp)
        assert worst > 0


def _settle_worker(rank, world, port, q):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        time.sleep(0.12 * rank)                              # the ranks' clocks start apart, as their set-up times do
        ran = []
        def step(i):                                         # every "window" is a collective: a rank running one alone would hang here
            t = torch.tensor([i], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            assert int(t.item()) == i * world                # ... and every rank is at the same window
            ran.append(i)
            time.sleep(0.002)
        n = sharded.settle_collectively(step, 60.0, torch.device("cpu"))
        q.put((rank, n, len(ran)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_settle_loop_runs_the_same_number_of_windows_on_every_rank():
    """the bench's settle loop decided from each rank's own wall clock whether to run eight more sharded windows — each a
    set of collectives — so a rank whose clock started earlier left for the barrier while another enqueued windows nobody joined.  The
    decision is a collective now: three gloo ranks whose clocks start 120 ms apart run the same number of windows (a multiple of 8), and at
    least as many as the slowest clock asks for."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_settle_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps: p.join(timeout=60)
    counts = {n for _, n, _ in got}
    assert len(counts) == 1 and all(n == m for _, n, m in got), got
    n = counts.pop()
    assert n % 8 == 0 and n >= 8
