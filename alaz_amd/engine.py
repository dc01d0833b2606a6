"""ctypes front end of libservicegraph.so (the C ABI of include/servicegraph.h).

This module is plumbing: it owns no algorithm.  Every result it returns was produced by the HIP
kernels behind the C ABI.  If the library is missing, or no gfx950 device is usable, it raises —
there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from .replay import EDGE_OUT_DTYPE, EVENT_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SG_LIB_PATH") or os.path.join(_HERE, "lib", "libservicegraph.so")   # SG_LIB_PATH: A/B runs of two builds on one box
# the development build (-DSG_DEV_KNOBS, alaz_amd/build.py): the only one that reads SG_* tuning knobs from the environment and has the
# SG_ABLATE bits / phase stamps compiled in.  ServiceGraph(dev_knobs=True) — tools/, the A/B tests of alternative kernel paths — loads it.
LIB_DEV_PATH = os.path.join(_HERE, "lib", "libservicegraph_dev.so")
#: the knobs the development build reads (servicegraph.hip sg_knob); ServiceGraph(dev_knobs=None) picks that build when one of them is set
DEV_KNOBS = ("SG_ABLATE", "SG_NP", "SG_HT", "SG_CT", "SG_NWG", "SG_NSUB", "SG_SPLIT", "SG_WARM", "SG_K1A", "SG_K1_NARROW", "SG_K1_LEGACY", "SG_K1B_U",
             "SG_K1B_THREADS", "SG_K1B_PACK", "SG_K1B_NO_ORDER", "SG_L2_GLOBAL", "SG_L2_U32", "SG_DH_G", "SG_K3_SLICES", "SG_K3_NO_FUSE", "SG_K4_FUSED", "SG_K5_GRID", "SG_K6_ONE_WG", "SG_DENSE_VALU",
             "SG_COPY_STREAMS", "SG_STAGE_SLOTS", "SG_ARENA")

SG_OK, SG_EINVAL, SG_ENOMEM, SG_ENODEV, SG_ENOSPC, SG_EAGAIN, SG_ESTATE = 0, -22, -12, -19, -28, -11, -71
F_IN, F_HID, F_EDGE = 32, 64, 8
STAT_SUM_WORDS, STAT_MAX_WORDS = 12, 2
REF_KNOWN, REF_LABEL, REF_OBIP = 0, 1, 2

#: every symbol include/servicegraph.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "sg_abi_version", "sg_weights_count", "sg_hash32", "sg_last_error", "sg_create", "sg_destroy",
    "sg_upsert_pod", "sg_delete_pod", "sg_upsert_service", "sg_delete_service", "sg_set_clock",
    "sg_set_label_count", "sg_load_weights", "sg_ingest", "sg_ingest_device", "sg_flush_window", "sg_flush_window_view", "sg_flush_begin", "sg_flush_end", "sg_flush_end_view",
    "sg_window_run", "sg_window_rows_buffer", "sg_window_close", "sg_window_obip_list",
    "sg_window_close_sharded", "sg_bind_buffers", "sg_window_features", "sg_window_layer", "sg_window_score", "sg_window_score_reset",
    "sg_window_read", "sg_window_reset", "sg_window_buffers", "sg_window_feat_buffer",
    "sg_halo_build", "sg_halo_pack", "sg_halo_unpack", "sg_window_close_gathered", "sg_halo_build_padded",
    "sg_halo_pack_padded", "sg_halo_unpack_padded", "sg_window_outbound_ips", "sg_stats_get",
    "sg_timing_enable", "sg_timing_reset", "sg_timing_get", "sg_timing_samples", "sg_timing_stride", "sg_latency_probe", "sg_set_warm", "sg_debug_stamps", "sg_route", "sg_window_hist", "sg_geometry_get",
    "sg_clock_probe", "sg_comm_probe", "sg_window_halo_counts", "sg_comm_unique_id", "sg_comm_create", "sg_comm_destroy", "sg_window_run_sharded", "sg_host_register", "sg_host_unregister", "sg_ingest_pinned", "sg_ingest_bulk",
]


class SgConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("device", C.c_int32), ("max_known_nodes", C.c_uint32),
                ("max_labels", C.c_uint32), ("max_outbound_ips", C.c_uint32), ("max_ips", C.c_uint32),
                ("max_edges", C.c_uint64), ("max_batch", C.c_uint32), ("layers", C.c_uint32),
                ("rank", C.c_uint32), ("world", C.c_uint32), ("k1_variant", C.c_uint32),
                ("max_window_events", C.c_uint64), ("windows_in_flight", C.c_uint32), ("max_alive", C.c_uint32), ("flags", C.c_uint32)]


CFG_EDGE_HISTOGRAM = 1
CFG_NO_WARM = 2
CFG_WARM = 4
ABI_VERSION = 6


def make_config(*, max_known_nodes: int, max_edges: int, layers: int = 1, max_labels: int = 256, max_outbound_ips: int = 64,
                max_ips: int = 0, max_batch: int = 1 << 16, device: int = 0, rank: int = 0, world: int = 1, k1_variant: int = 0,
                max_window_events: int = 0, windows_in_flight: int = 1, max_alive: int = 0, flags: int = 0) -> "SgConfig":
    """sg_config by field name (struct_size and abi_version filled in)."""
    return SgConfig(C.sizeof(SgConfig), ABI_VERSION, device, max_known_nodes, max_labels, max_outbound_ips, max_ips or max_known_nodes,
                    max_edges, max_batch, layers, rank, world, k1_variant, max_window_events, windows_in_flight, max_alive, flags)


class SgGeometry(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("k1_variant", "k1_narrow", "partitions", "table_slots", "pass_a_workgroups", "cache_slots",
                                          "join_l2_in_lds", "tile_records", "endpoint_bits", "piece_bytes", "pass_b_split", "pass_a_teams", "warm_windows")]


class SgStats(C.Structure):
    _fields_ = [("events_in", C.c_uint64), ("events_dropped_src", C.c_uint64), ("events_dropped_ring", C.c_uint64),
                ("events_dropped_cap", C.c_uint64), ("windows", C.c_uint64), ("last_window_events", C.c_uint64),
                ("last_window_edges", C.c_uint64), ("last_window_nodes", C.c_uint64),
                ("last_window_tmin_ms", C.c_int64), ("last_window_tmax_ms", C.c_int64), ("h2d_bytes", C.c_uint64),
                ("events_misrouted", C.c_uint64), ("halo_overflow", C.c_uint64),
                ("alive_in", C.c_uint64), ("alive_dropped", C.c_uint64),
                ("join_word_updates", C.c_uint64), ("join_full_uploads", C.c_uint64), ("ingest_waits", C.c_uint64),
                ("windows_warm", C.c_uint64), ("windows_cold", C.c_uint64),
                ("windows_delta", C.c_uint64), ("windows_plain", C.c_uint64), ("last_window_new_edges", C.c_uint64)]


class ServiceGraphError(RuntimeError):
    def __init__(self, rc: int, msg: str):
        super().__init__(f"servicegraph rc={rc}: {msg}")
        self.rc = rc


_lib = None
_lib_dev = None


def load_library(path: str = LIB_PATH, dev: bool = False) -> C.CDLL:
    """dlopen the engine.  torch is imported first so that the HIP runtime torch ships is the one
    both share (same soname, one copy per process): device pointers of torch tensors are then
    valid arguments of sg_ingest_device / the halo calls."""
    global _lib, _lib_dev
    if dev:
        if _lib_dev is not None:
            return _lib_dev
        path = os.environ.get("SG_LIB_DEV", LIB_DEV_PATH)   # (another development build: A/B runs of two kernel forms on one box)
    else:
        if _lib is not None:
            return _lib
        path = os.environ.get("SG_LIB", path)          # another build of the same sources
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m alaz_amd.build` "
                           "(hipcc, gfx950). The ServiceGraph engine has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads libamdhip64 with RTLD_GLOBAL semantics first)
    except Exception:
        pass
    # (the development build is loaded locally: it exports the same names as the shipped library and both may be loaded in one process;
    # both are linked -Bsymbolic, so neither's own calls can land in the other)
    lib = C.CDLL(path, mode=C.RTLD_LOCAL if dev else C.RTLD_GLOBAL)
    H, P = C.c_void_p, C.c_void_p
    u32, u64, sz = C.c_uint32, C.c_uint64, C.c_size_t
    sig = {
        "sg_abi_version": (u32, []), "sg_weights_count": (sz, [u32]), "sg_hash32": (u32, [u32]),
        "sg_last_error": (C.c_char_p, [H]),
        "sg_create": (C.c_int, [C.POINTER(SgConfig), C.POINTER(H)]), "sg_destroy": (C.c_int, [H]),
        "sg_upsert_pod": (C.c_int, [H, u32, u32]), "sg_delete_pod": (C.c_int, [H, u32]),
        "sg_upsert_service": (C.c_int, [H, u32, u32]), "sg_delete_service": (C.c_int, [H, u32]),
        "sg_set_clock": (C.c_int, [H, u64, u64]), "sg_set_label_count": (C.c_int, [H, u32]),
        "sg_load_weights": (C.c_int, [H, P, sz]),
        "sg_ingest": (C.c_int, [H, P, sz]), "sg_ingest_device": (C.c_int, [H, P, sz, P]),
        "sg_flush_window": (C.c_int, [H, u64, P, sz, C.POINTER(sz)]),
        "sg_flush_window_view": (C.c_int, [H, u64, C.POINTER(C.c_void_p), C.POINTER(sz)]),
        "sg_flush_begin": (C.c_int, [H, u64]),
        "sg_flush_end": (C.c_int, [H, C.c_void_p, sz, C.POINTER(sz)]),
        "sg_flush_end_view": (C.c_int, [H, C.POINTER(C.c_void_p), C.POINTER(sz)]),
        "sg_window_run": (C.c_int, [H, P]), "sg_window_rows_buffer": (C.c_int, [H, C.POINTER(P)]),
        "sg_window_close": (C.c_int, [H, P]),
        "sg_window_obip_list": (C.c_int, [H, P, u32, P, P]),
        "sg_bind_buffers": (C.c_int, [H, P, P, C.POINTER(P), u32]),
        "sg_window_close_sharded": (C.c_int, [H, P, P, P]),
        "sg_window_features": (C.c_int, [H, P]), "sg_window_layer": (C.c_int, [H, u32, P]),
        "sg_window_score": (C.c_int, [H, P]), "sg_window_score_reset": (C.c_int, [H, P]), "sg_window_read": (C.c_int, [H, P, sz, C.POINTER(sz)]),
        "sg_window_reset": (C.c_int, [H, P]),
        "sg_window_buffers": (C.c_int, [H, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(sz)]),
        "sg_window_feat_buffer": (C.c_int, [H, u32, C.POINTER(P), C.POINTER(sz)]),
        "sg_halo_build": (C.c_int, [H, P, u32, P, P]), "sg_halo_pack": (C.c_int, [H, u32, P, u32, P, P]),
        "sg_halo_unpack": (C.c_int, [H, u32, P, u32, P, P]),
        "sg_window_close_gathered": (C.c_int, [H, P, u32, u32, P]),
        "sg_halo_build_padded": (C.c_int, [H, P, u32, P]), "sg_halo_pack_padded": (C.c_int, [H, u32, P, u32, P, P]),
        "sg_halo_unpack_padded": (C.c_int, [H, u32, P, u32, P, P]),
        "sg_window_outbound_ips": (C.c_int, [H, P, sz, C.POINTER(sz)]),
        "sg_stats_get": (C.c_int, [H, C.POINTER(SgStats)]),
        "sg_timing_enable": (C.c_int, [H, C.c_int]), "sg_timing_reset": (C.c_int, [H]),
        "sg_timing_get": (C.c_int, [H, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]),
        "sg_set_warm": (C.c_int, [H, C.c_int]), "sg_timing_stride": (C.c_int, [H, C.c_uint32]),
        "sg_timing_samples": (C.c_int, [H, C.c_int, C.POINTER(C.c_double), sz, C.POINTER(sz)]),
        "sg_latency_probe": (C.c_int, [H, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_double)]),
        "sg_debug_stamps": (C.c_int, [H, P, sz]),
        "sg_clock_probe": (C.c_int, [H, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "sg_comm_probe": (C.c_int, []), "sg_window_halo_counts": (C.c_int, [H, P, sz]),
        "sg_route": (C.c_int, [H, P, sz, u32, P]),
        "sg_window_hist": (C.c_int, [H, P, sz, C.POINTER(sz)]),
        "sg_geometry_get": (C.c_int, [H, C.POINTER(SgGeometry)]),
        "sg_comm_unique_id": (C.c_int, [P, sz]), "sg_comm_create": (C.c_int, [P, sz, C.c_int, C.c_int, C.c_int, C.POINTER(P)]),
        "sg_comm_destroy": (C.c_int, [P]), "sg_window_run_sharded": (C.c_int, [H, P, P]),
        "sg_host_register": (C.c_int, [H, P, sz]), "sg_host_unregister": (C.c_int, [H, P]), "sg_ingest_pinned": (C.c_int, [H, P, sz]),
        "sg_ingest_bulk": (C.c_int, [H, P, sz, C.c_int, C.POINTER(u64)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)          # AttributeError if the library does not export it
        f.restype = res; f.argtypes = args
    if dev:
        _lib_dev = lib
    else:
        _lib = lib
    return lib


class RcclComm:
    """The engine library's own RCCL communicator (sg_comm_*): rank 0 draws the unique id, `bcast(bytes) -> bytes` hands it to
    every rank (torch.distributed.broadcast_object_list, MPI, a file ...), every rank joins."""

    @staticmethod
    def probe() -> bool:
        """Can the library reach RCCL in this process?  (Agree on a fallback with every rank BEFORE constructing: a rank that fails
        inside the constructor leaves the others waiting in ncclCommInitRank.)"""
        return load_library().sg_comm_probe() == SG_OK

    def __init__(self, rank: int, world: int, device: int, bcast):
        l = load_library()
        buf = (C.c_char * 128)()
        rc = l.sg_comm_unique_id(buf, 128) if rank == 0 else SG_OK
        raw = bcast(bytes(buf.raw) if rc == SG_OK else b"")      # an empty id tells every rank that rank 0 has no RCCL: all of them raise
        if not raw:
            raise ServiceGraphError(rc if rc != SG_OK else SG_ENODEV, "sg_comm_unique_id: librccl could not be loaded on rank 0")
        idb = (C.c_char * 128).from_buffer_copy(raw)
        p = C.c_void_p()
        rc = l.sg_comm_create(idb, 128, rank, world, device, C.byref(p))
        if rc != SG_OK:
            raise ServiceGraphError(rc, "sg_comm_create (ncclCommInitRank) failed")
        self._l, self.ptr, self.rank, self.world = l, p, rank, world

    def close(self):
        if getattr(self, "ptr", None):
            self._l.sg_comm_destroy(self.ptr); self.ptr = None


def ip_u32(s: str) -> int:
    a, b, c, d = (int(x) for x in s.split("."))
    return (a << 24) | (b << 16) | (c << 8) | d


class ServiceGraph:
    """One engine handle (one GPU / one shard)."""

    def __init__(self, *, max_known_nodes: int, max_edges: int, layers: int = 1, max_labels: int = 1024,
                 max_outbound_ips: int = 1024, max_ips: int = 0, max_batch: int = 1 << 20, device: int = 0,
                 rank: int = 0, world: int = 1, k1_variant: int = 0, max_window_events: int = 0, windows_in_flight: int = 1,
                 edge_histogram: bool = False, warm: Optional[bool] = None, dev_knobs: Optional[bool] = None):
        # dev_knobs: the development build (SG_* environment knobs, SG_ABLATE, phase stamps).  None = that build when a knob is set in the
        # environment (tools/k1_sweep.py, tools/stamps.py, the A/B tests of alternative kernel paths), the shipped library otherwise.
        if dev_knobs is None:
            dev_knobs = any(k in os.environ for k in DEV_KNOBS)
        self._l = load_library(dev=dev_knobs)
        cfg = make_config(max_known_nodes=max_known_nodes, max_edges=max_edges, layers=layers, max_labels=max_labels,
                          max_outbound_ips=max_outbound_ips, max_ips=max_ips, max_batch=max_batch, device=device, rank=rank, world=world,
                          k1_variant=k1_variant, max_window_events=max_window_events, windows_in_flight=windows_in_flight,
                          flags=(CFG_EDGE_HISTOGRAM if edge_histogram else 0) | (0 if warm is None else (CFG_WARM if warm else CFG_NO_WARM)))   # warm: None = the engine's own rule
        h = C.c_void_p()
        rc = self._l.sg_create(C.byref(cfg), C.byref(h))
        if rc != SG_OK:
            raise ServiceGraphError(rc, "sg_create failed (no usable gfx950 device, or bad config); there is no CPU fallback")
        self._h = h
        self.layers = layers
        self.max_edges = max_edges
        self.max_batch = max_batch
        self.rank, self.world = rank, world

    # ---- plumbing ----
    def _ck(self, rc: int, allow=()):
        if rc != SG_OK and rc not in allow:
            raise ServiceGraphError(rc, (self._l.sg_last_error(self._h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._l.sg_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def geometry(self) -> dict:
        """What sg_create chose for K1 (sg_geometry_get)."""
        g = SgGeometry()
        self._ck(self._l.sg_geometry_get(self._h, C.byref(g)))
        return {n: int(getattr(g, n)) for n, _ in SgGeometry._fields_}

    def k1_kernels(self) -> tuple:
        """Names of the two K1 kernels this engine launches (as rocprofv3 lists them), or the single global-table kernel."""
        g = self.geometry()
        if g["k1_variant"] == 1: return ("k1_resolve_aggregate",)
        if not g["k1_narrow"]: return ("k1a_partition", "k1b_merge")
        return ("k1a_team_partition" if g["pass_a_teams"] else "k1a_tile_partition", "k1b_stream_merge")

    # ---- join tables (aggregator/persist.go:55-71,114-130) ----
    def upsert_pod(self, ip: int, node_id: int): self._ck(self._l.sg_upsert_pod(self._h, ip, node_id))
    def delete_pod(self, ip: int): self._ck(self._l.sg_delete_pod(self._h, ip))
    def upsert_service(self, ip: int, node_id: int): self._ck(self._l.sg_upsert_service(self._h, ip, node_id))
    def delete_service(self, ip: int): self._ck(self._l.sg_delete_service(self._h, ip))
    def set_clock(self, first_kernel_ns: int, first_user_ns: int): self._ck(self._l.sg_set_clock(self._h, first_kernel_ns, first_user_ns))
    def set_label_count(self, n: int): self._ck(self._l.sg_set_label_count(self._h, n))

    def load_weights(self, w: np.ndarray):
        w = np.ascontiguousarray(w, dtype=np.float32)
        self._ck(self._l.sg_load_weights(self._h, w.ctypes.data, len(w)))

    # ---- ingest ----
    def ingest(self, events: np.ndarray) -> int:
        ev = np.ascontiguousarray(events)
        assert ev.dtype == EVENT_DTYPE
        return self._ck(self._l.sg_ingest(self._h, ev.ctypes.data, len(ev)), allow=(SG_EAGAIN,))

    def host_register(self, arr: np.ndarray):
        """Page-lock a (contiguous) numpy array so that ingest_pinned can read slices of it without the staging copy."""
        assert arr.flags.c_contiguous
        self._ck(self._l.sg_host_register(self._h, arr.ctypes.data, arr.nbytes))

    def host_unregister(self, arr: np.ndarray): self._ck(self._l.sg_host_unregister(self._h, arr.ctypes.data))

    def ingest_pinned(self, events: np.ndarray) -> int:
        """events: a slice of a registered array; it must stay unchanged until the window has been closed."""
        assert events.dtype == EVENT_DTYPE and events.flags.c_contiguous
        return self._ck(self._l.sg_ingest_pinned(self._h, events.ctypes.data, len(events)), allow=(SG_EAGAIN,))

    def ingest_bulk(self, events: np.ndarray, pinned: bool = False) -> int:
        """All of `events` in max_batch pieces, waiting for a free staging slot instead of dropping; returns the waits."""
        assert events.dtype == EVENT_DTYPE and events.flags.c_contiguous
        w = C.c_uint64(0)
        self._ck(self._l.sg_ingest_bulk(self._h, events.ctypes.data, len(events), 1 if pinned else 0, C.byref(w)))
        return int(w.value)

    def ingest_device(self, dev_ptr: int, n: int, stream: int = 0):
        self._ck(self._l.sg_ingest_device(self._h, dev_ptr, n, stream or None))

    # ---- window ----
    def flush_window(self, window_end_ms: int = 0, cap: Optional[int] = None) -> np.ndarray:
        cap = self.max_edges if cap is None else cap
        out = np.zeros(cap, dtype=EDGE_OUT_DTYPE)
        n = C.c_size_t(0)
        self._ck(self._l.sg_flush_window(self._h, window_end_ms, out.ctypes.data, cap, C.byref(n)))
        return out[: min(n.value, cap)]

    def flush_window_view(self, window_end_ms: int = 0) -> np.ndarray:
        """The window's rows as a read-only VIEW of the engine's page-locked host buffer (no copy): valid until the next
        flush_window / flush_window_view / window_read on this engine."""
        ptr = C.c_void_p(); n = C.c_size_t(0)
        self._ck(self._l.sg_flush_window_view(self._h, window_end_ms, C.byref(ptr), C.byref(n)))
        return self._rows_view(ptr, n)

    def flush_begin(self, window_end_ms: int = 0):
        """Close the window and enqueue its pipeline (sg_flush_begin); ingest calls from here on fill the next window."""
        self._ck(self._l.sg_flush_begin(self._h, window_end_ms))

    def flush_end_view(self) -> np.ndarray:
        """The rows of the window flush_begin closed, as flush_window_view returns them (any thread; the engine lock is not held
        while the rows are fetched)."""
        ptr = C.c_void_p(); n = C.c_size_t(0)
        self._ck(self._l.sg_flush_end_view(self._h, C.byref(ptr), C.byref(n)))
        return self._rows_view(ptr, n)

    def flush_end(self, cap: int | None = None) -> np.ndarray:
        cap = self.max_edges if cap is None else cap
        out = np.zeros(cap, dtype=EDGE_OUT_DTYPE); n = C.c_size_t(0)
        self._ck(self._l.sg_flush_end(self._h, out.ctypes.data, cap, C.byref(n)))
        return out[: min(n.value, cap)]

    @staticmethod
    def _rows_view(ptr, n) -> np.ndarray:
        if n.value == 0:
            z = np.zeros(0, dtype=EDGE_OUT_DTYPE); z.flags.writeable = False
            return z
        buf = (C.c_char * (n.value * EDGE_OUT_DTYPE.itemsize)).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=EDGE_OUT_DTYPE)
        a.flags.writeable = False
        return a

    def window_run(self, stream: int = 0): self._ck(self._l.sg_window_run(self._h, stream or None))
    def window_close(self, stream: int = 0): self._ck(self._l.sg_window_close(self._h, stream or None))
    def window_features(self, stream: int = 0): self._ck(self._l.sg_window_features(self._h, stream or None))
    def window_layer(self, l: int, stream: int = 0): self._ck(self._l.sg_window_layer(self._h, l, stream or None))
    def window_score(self, stream: int = 0): self._ck(self._l.sg_window_score(self._h, stream or None))

    def window_score_reset(self, stream: int = 0): self._ck(self._l.sg_window_score_reset(self._h, stream or None))

    def window_run_sharded(self, comm: "RcclComm", stream: int = 0):
        """K1 pass B .. K5 of this shard's window with every exchange, ONE C call (sg_window_run_sharded); rows stay on the device."""
        self._ck(self._l.sg_window_run_sharded(self._h, comm.ptr, stream or None))
    def window_reset(self, stream: int = 0): self._ck(self._l.sg_window_reset(self._h, stream or None))

    def halo_counts(self, world: int) -> np.ndarray:
        out = np.zeros(world, dtype=np.uint32)
        self._ck(self._l.sg_window_halo_counts(self._h, out.ctypes.data, world))
        return out

    def window_close_sharded(self, d_union_ips: int, d_union_n: int, stream: int = 0):
        self._ck(self._l.sg_window_close_sharded(self._h, d_union_ips, d_union_n, stream or None))

    def window_obip_list(self, d_list: int, cap: int, d_n: int, stream: int = 0):
        self._ck(self._l.sg_window_obip_list(self._h, d_list, cap, d_n, stream or None))

    def bind_buffers(self, stats_sum: int, stats_max: int, feat_rows):
        arr = (C.c_void_p * max(1, len(feat_rows)))(*feat_rows)
        self._ck(self._l.sg_bind_buffers(self._h, stats_sum, stats_max, arr, len(feat_rows)))

    def window_read(self, cap: Optional[int] = None) -> np.ndarray:
        cap = self.max_edges if cap is None else cap
        out = np.zeros(cap, dtype=EDGE_OUT_DTYPE)
        n = C.c_size_t(0)
        self._ck(self._l.sg_window_read(self._h, out.ctypes.data, cap, C.byref(n)))
        return out[: min(n.value, cap)]

    def window_buffers(self):
        a, b, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._ck(self._l.sg_window_buffers(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def feat_buffer(self, l: int):
        p, w = C.c_void_p(), C.c_size_t()
        self._ck(self._l.sg_window_feat_buffer(self._h, l, C.byref(p), C.byref(w)))
        return p.value, w.value

    def rows_buffer(self) -> int:
        p = C.c_void_p()
        self._ck(self._l.sg_window_rows_buffer(self._h, C.byref(p)))
        return p.value

    def halo_build(self, d_ids: int, cap: int, d_counts: int, stream: int = 0): self._ck(self._l.sg_halo_build(self._h, d_ids, cap, d_counts, stream or None))
    def halo_pack(self, l: int, d_ids: int, n: int, d_rows: int, stream: int = 0): self._ck(self._l.sg_halo_pack(self._h, l, d_ids, n, d_rows, stream or None))
    def halo_unpack(self, l: int, d_ids: int, n: int, d_rows: int, stream: int = 0): self._ck(self._l.sg_halo_unpack(self._h, l, d_ids, n, d_rows, stream or None))

    def window_close_gathered(self, d_gathered: int, stride: int, world: int, stream: int = 0):
        self._ck(self._l.sg_window_close_gathered(self._h, d_gathered, stride, world, stream or None))

    def halo_build_padded(self, d_req: int, capp: int, stream: int = 0): self._ck(self._l.sg_halo_build_padded(self._h, d_req, capp, stream or None))
    def halo_pack_padded(self, l: int, d_serve: int, capp: int, d_rows: int, stream: int = 0): self._ck(self._l.sg_halo_pack_padded(self._h, l, d_serve, capp, d_rows, stream or None))
    def halo_unpack_padded(self, l: int, d_req: int, capp: int, d_rows: int, stream: int = 0): self._ck(self._l.sg_halo_unpack_padded(self._h, l, d_req, capp, d_rows, stream or None))

    def outbound_ips(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._ck(self._l.sg_window_outbound_ips(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            self._ck(self._l.sg_window_outbound_ips(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def window_hist(self) -> np.ndarray:
        """[rows][16] u32 latency histogram bins of the last read window (engine created with edge_histogram=True)."""
        n = C.c_size_t(0)
        self._ck(self._l.sg_window_hist(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 16), dtype=np.uint32)
        if n.value:
            self._ck(self._l.sg_window_hist(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def stats(self) -> SgStats:
        s = SgStats()
        self._ck(self._l.sg_stats_get(self._h, C.byref(s)))
        return s

    def timing_enable(self, mask: int = 1): self._ck(self._l.sg_timing_enable(self._h, int(mask)))
    def timing_reset(self): self._ck(self._l.sg_timing_reset(self._h))
    def timing_stride(self, n: int): self._ck(self._l.sg_timing_stride(self._h, int(n)))

    def timing(self, kernel: int) -> Tuple[float, int]:
        us, n = C.c_double(), C.c_uint64()
        self._ck(self._l.sg_timing_get(self._h, kernel, C.byref(us), C.byref(n)))
        return us.value, n.value

    def set_warm(self, on: bool = True):
        """warm windows on / off at run time (off: every window is rebuilt from nothing; the rows are the same either way)"""
        self._ck(self._l.sg_set_warm(self._h, 1 if on else 0))

    def timing_samples(self, kernel: int, cap: int = 4096) -> np.ndarray:
        """every record of a timing group since timing_reset(), microseconds, in launch order"""
        out = np.zeros(cap, dtype=np.float64); n = C.c_size_t()
        self._ck(self._l.sg_timing_samples(self._h, kernel, out.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n)))
        return out[: min(cap, n.value)]

    def latency_probe(self, nbytes: int, steps: int, warm: bool = False, loaded: bool = False) -> float:
        """ns per dependent load through `nbytes` of device memory (HBM: far beyond the Infinity Cache, cold; L2: 2 MiB, warm);
        loaded: 65 536 chains in flight at once, one of them timed"""
        ns = C.c_double()
        self._ck(self._l.sg_latency_probe(self._h, int(nbytes), int(steps), (1 if warm else 0) | (2 if loaded else 0), C.byref(ns)))
        return ns.value

    def clock_probe(self, spin_us: int = 200) -> Tuple[float, float]:
        """(MHz under an all-CU spin launched now, MHz averaged over the pass-A launches since the last call)."""
        a, b = C.c_double(), C.c_double()
        self._ck(self._l.sg_clock_probe(self._h, int(spin_us), C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_stamps(self) -> np.ndarray:
        """[kernel 0..3][workgroup][8] phase stamps (100 MHz ticks); zeros unless SG_ABLATE & 0x100."""
        out = np.zeros((4, 4096, 8), dtype=np.uint64)
        self._ck(self._l.sg_debug_stamps(self._h, out.ctypes.data, out.size))
        return out

    def route(self, events: np.ndarray, world: int) -> np.ndarray:
        ev = np.ascontiguousarray(events)
        out = np.zeros(len(ev), dtype=np.uint32)
        self._ck(self._l.sg_route(self._h, ev.ctypes.data, len(ev), world, out.ctypes.data))
        return out
