// sg_kernels.h — hand-written gfx950 kernels of the ServiceGraph engine (K1..K6).
// Included once by servicegraph.hip.  Every kernel is HBM/L2-bound integer or gather work except
// the per-node dense blocks of K4/K5, which run on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32).
#pragma once
#include "sg_device.h"
#include "sg_hash.h"

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 hash_key64(u64 k) { return sg_fmix32((u32)k ^ sg_fmix32((u32)(k >> 32) + 0x9e3779b9u)); }

// owner shard of a node: by its stable ref; OBIP nodes (window-local indices) by their IP.
__host__ __device__ __forceinline__ u32 owner_hash_ref(u32 ref) { return sg_fmix32(ref); }
__host__ __device__ __forceinline__ u32 owner_hash_obip(u32 ip) { return sg_fmix32(ip ^ 0xA5A5F00Du); }


// two independent bucket reads; t may point to LDS (staged copy) or to global memory
__device__ __forceinline__ u64 ip_probe(const u64* t, u32 mask, u32 ip) {
    const u32 bmask = mask >> 1;
    const ulonglong2 a = reinterpret_cast<const ulonglong2*>(t)[ip_h1(ip, bmask)];
    const ulonglong2 b = reinterpret_cast<const ulonglong2*>(t)[ip_h2(ip, bmask)];
    u64 e = SG_IP_EMPTY;
    e = ((u32)a.x == ip && a.x != SG_IP_EMPTY) ? a.x : e;
    e = ((u32)a.y == ip && a.y != SG_IP_EMPTY) ? a.y : e;
    e = ((u32)b.x == ip && b.x != SG_IP_EMPTY) ? b.x : e;
    e = ((u32)b.y == ip && b.y != SG_IP_EMPTY) ? b.y : e;
    return e;
}

// A fresh read of an LDS word other lanes may be writing.  (Not `volatile`: a volatile access through a pointer the
// compiler has to infer the address space of stays a FLAT load — flat_load_dwordx2 sc0 sc1 plus s_waitcnt vmcnt(0),
// which also drains every global store in flight.)
__device__ __forceinline__ u64 lds_fresh_u64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ u32 lds_fresh_u32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// find-or-insert in an open-addressing u64 key table.  A plain load may return a stale EMPTY (the
// XCD L2s are not coherent); every EMPTY observation is confirmed by the device-scope CAS, and a
// slot never changes once it holds a key, so a non-EMPTY observation is always final.
__device__ __forceinline__ bool table_slot(u64* keys, u32 mask, u64 key, u64 empty, u32 h, u32& slot) {
    for (u32 p = 0; p <= mask; ++p) {
        u64 k = keys[h];
        if (k == empty) {
            k = atomicCAS(&keys[h], empty, key);
            if (k == empty) { slot = h; return true; }
        }
        if (k == key) { slot = h; return true; }
        h = (h + 1) & mask;
    }
    return false;
}

// Wave-wide reductions without LDS traffic: an xor butterfly inside each row of 16 lanes with DPP
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row results
// are combined through v_readlane.  All 64 lanes must be active; every lane gets the result.
// (A __shfl_xor chain is six dependent ds_bpermute round trips per value: seven values per wave at
// the end of k1a_partition cost ~1.5 us that way.)
template <int CTRL> __device__ __forceinline__ u32 dpp32(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
// (bound_ctrl:1 — no lane of these controls reads out of bounds, but it tells the compiler the old value is dead: no v_mov 0 + hazard nops per use)
template <int CTRL> __device__ __forceinline__ u32 dpp32b(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ u64 dpp64(u64 v) { return (u64)dpp32<CTRL>((u32)v) | ((u64)dpp32<CTRL>((u32)(v >> 32)) << 32); }
__device__ __forceinline__ u32 rdlane32(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 rdlane64(u64 v, int l) { return (u64)rdlane32((u32)v, l) | ((u64)rdlane32((u32)(v >> 32), l) << 32); }
#define SG_WAVE_REDUCE(T, DPP, RD, OP)                                                                   \
    { T o;                                                                                                \
      o = DPP<0xB1>(v); v = OP(v, o); o = DPP<0x4E>(v); v = OP(v, o);                                     \
      o = DPP<0x141>(v); v = OP(v, o); o = DPP<0x140>(v); v = OP(v, o);                                   \
      const T r0 = RD(v, 0), r1 = RD(v, 16), r2 = RD(v, 32), r3 = RD(v, 48);                              \
      return OP(OP(r0, r1), OP(r2, r3)); }
// xor-butterfly partner of a lane without LDS: strides 1, 2 = quad_perm; 4 = row_half_mirror then a quad reverse
// ((i ^ 7) ^ 3 = i ^ 4); 8 = row_ror:8; 16 / 32 = the gfx950 v_permlane16_swap / v_permlane32_swap (both operands
// hold v: afterwards one result holds the lower member of every pair in both places, the other the upper one).
__device__ __forceinline__ float xor_partner_f32(float x, int stride) {
    const u32 v = __float_as_uint(x);
    u32 o;
    switch (stride) {
        case 1:  o = dpp32<0xB1>(v); break;
        case 2:  o = dpp32<0x4E>(v); break;
        case 4:  o = dpp32<0x1B>(dpp32<0x141>(v)); break;
        case 8:  o = dpp32<0x128>(v); break;
        case 16: { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); o = (threadIdx.x & 16u) ? r[0] : r[1]; break; }
        default: { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); o = (threadIdx.x & 32u) ? r[0] : r[1]; break; }
    }
    return __uint_as_float(o);
}
// r_l <- sum over the wave in the canonical butterfly order (strides 32, 16, 8, 4, 2, 1): what the oracle defines
// for the score head ("xor-butterfly sum"); every lane ends with the same bits
__device__ __forceinline__ float wave_butterfly_sum_f32(float r) {
    r = r + xor_partner_f32(r, 32); r = r + xor_partner_f32(r, 16); r = r + xor_partner_f32(r, 8);
    r = r + xor_partner_f32(r, 4);  r = r + xor_partner_f32(r, 2);  r = r + xor_partner_f32(r, 1);
    return r;
}
// Storage order of the score head's per-node projections P, Q (private to K4's fused projection, k5_node_proj and
// k5_edge_score): hidden unit j lives at position (j & 15) * 4 + (j >> 4), so that the four units {q, q+16, q+32, q+48}
// a lane of k5_edge_score owns are one 16-byte load.
#define SG_PQ_POS(j) ((((j) & 15) << 2) | ((j) >> 4))
#define SG_MEAN_BLOCK 512        // neighbours per block of the canonical mean: block sums are added in block order
#define SG_OP_MIN(a, b) ((b) < (a) ? (b) : (a))
#define SG_OP_MAX(a, b) ((b) > (a) ? (b) : (a))
#define SG_OP_ADD(a, b) ((a) + (b))
__device__ __forceinline__ u64 wave_min_u64(u64 v) SG_WAVE_REDUCE(u64, dpp64, rdlane64, SG_OP_MIN)
__device__ __forceinline__ u64 wave_max_u64(u64 v) SG_WAVE_REDUCE(u64, dpp64, rdlane64, SG_OP_MAX)
__device__ __forceinline__ u64 wave_sum_u64(u64 v) SG_WAVE_REDUCE(u64, dpp64, rdlane64, SG_OP_ADD)
__device__ __forceinline__ u32 wave_sum_u32(u32 v) SG_WAVE_REDUCE(u32, dpp32, rdlane32, SG_OP_ADD)

// "error" classification: HTTP/HTTP2 >= 500; POSTGRES/REDIS/MYSQL == 2 (ebpf/c/postgres.c:91,
// redis.c:10, mysql.c:36).
__device__ __forceinline__ u32 is_error(u32 proto, u32 status) {
    const bool http = (proto == SG_PROTO_HTTP) | (proto == SG_PROTO_HTTP2);
    const bool sql = (proto == SG_PROTO_POSTGRES) | (proto == SG_PROTO_REDIS) | (proto == SG_PROTO_MYSQL);
    return (http & (status >= 500u)) | (sql & (status == 2u)) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------------
// block-level primitives (single-workgroup kernels use 1024 threads = 16 waves)
// ------------------------------------------------------------------------------------------------
// exclusive scan of one value per thread over the workgroup; *total = sum.  wsum: LDS [NT/64 + 1].
template <int NT>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* wsum, u32* total) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const u32 o = __shfl_up(incl, s, 64); if ((int)lane >= s) incl += o; }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const u32 x = lane < NT / 64 ? wsum[lane] : 0u;
        u32 xi = x;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const u32 o = __shfl_up(xi, s, 64); if ((int)lane >= s) xi += o; }
        if (lane < NT / 64) wsum[lane] = xi - x;
        if (lane == 63) wsum[NT / 64] = xi;
    }
    __syncthreads();
    *total = wsum[NT / 64];
    return wsum[wave] + incl - v;
}

// ---- canonical node numbering ---------------------------------------------------------------------
__device__ __forceinline__ u32 lower_bound_u32(const u32* a, u32 n, u32 v) {
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo;
}

// canonical dense index of a node ref: KNOWN ids, then LABEL ids, then OBIP by ascending IP.
__device__ __forceinline__ u32 dense_of(const Dev& d, u32 ref, u32 nk, u32 nl, u32 nob) {
    const u32 t = SG_REF_TYPE(ref), v = SG_REF_VALUE(ref);
    if (t == SG_REF_KNOWN) return v;
    if (t == SG_REF_LABEL) return nk + v;
    const u32 ip = (u32)d.obkeys[v];
    const u32 r = lower_bound_u32(d.ob_sorted, nob, ip);
    return (r < nob && d.ob_sorted[r] == ip) ? nk + nl + r : SG_NONE;   // not listed: more raw outbound IPs than max_outbound_ips
}
__device__ __forceinline__ u32 ref_of_dense(u32 v, u32 nk, u32 nl) {
    if (v < nk) return SG_MAKE_REF(SG_REF_KNOWN, v);
    if (v < nk + nl) return SG_MAKE_REF(SG_REF_LABEL, v - nk);
    return SG_MAKE_REF(SG_REF_OBIP, v - nk - nl);
}
__device__ __forceinline__ u32 owner_of_dense(const Dev& d, u32 v, u32 nk, u32 nl) {
    if (v < nk + nl) return owner_hash_ref(ref_of_dense(v, nk, nl)) % d.world;
    return owner_hash_obip(d.ob_sorted[v - nk - nl]) % d.world;
}

// ------------------------------------------------------------------------------------------------
// K1  resolve_aggregate: events -> (from, to) node refs -> per-edge integer accumulators.
// Replaces extractAddressPair + setFromToV2 + ReverseDirection + the per-request PersistRequest
// (aggregator/data.go:1760-1767, 827-870; datastore/dto.go:226-231; backend.go:819-847).
// Algorithmic bytes: 32 per event read + 32 per distinct edge written.
// ------------------------------------------------------------------------------------------------
struct K1Local { u64 tmin, tmax; u32 maxlabel, dsrc, dcap, misr, acc, lost; };   // lost: accepted by a lane of this workgroup, then dropped for capacity
struct K1Ev { u64 key, dur, wt; u32 err; u32 alive; };
#define SG_DUR_MAX ((1ull << 62) - 1)      // durations saturate here: bits 62/63 of a single record carry flags

// f-3: latency histogram bin of a duration (include/servicegraph.h): 0 below 2^17 ns, one bin per octave, 15 from 2^31 ns
__device__ __forceinline__ u32 hist_bin32(u32 dur) { return dur < (1u << 17) ? 0u : (dur >> 31) ? 15u : (15u - (u32)__builtin_clz(dur)); }   // 31 - clz - 16
__device__ __forceinline__ u32 hist_bin64(u64 dur) { return (dur >> 32) ? 15u : hist_bin32((u32)dur); }

// ---- the join, general form: block table (global copy), then the cuckoo table.  Returns kind << 30 | id, 0 = unknown IP;
// kind 3 = the IP is in both reference maps (id = the service; the pod id is in the small second table).
__device__ __forceinline__ u32 join_general(const Dev& d, u32 ip) {
    if (d.jl2_words) {
        const u32 b = ip >> 8;
        const u64 e1 = d.jl1[jl1_h1(b, d.jl1mask)], e2 = d.jl1[jl1_h2(b, d.jl1mask)];
        const u32 blk = (u32)e1 == b ? (u32)(e1 >> 32) : ((u32)e2 == b ? (u32)(e2 >> 32) : 0u);
        const u32 v = d.jl2[(blk << 8) | (ip & 255u)];
        if (v) return v;
    }
    if (d.ck_n) { const u64 e = ip_probe(d.iptab, d.ipmask, ip); if (e != SG_IP_EMPTY) return (u32)(e >> 32); }
    return 0u;
}
__device__ __forceinline__ void join_pod_svc(const Dev& d, u32 ip, u32& pod, u32& svc) {
    const u32 v = join_general(d, ip), kind = v >> 30, id = v & 0x3FFFFFFFu;
    if (kind == 1) pod = id;
    else if (kind == 2) svc = id;
    else if (kind == 3) { svc = id; const u64 e2 = ip_probe(d.iptab2, d.ipmask2, ip); if (e2 != SG_IP_EMPTY) pod = (u32)(e2 >> 32) & 0x3FFFFFFFu; }
}

// the join of one event, general form: edge key, or a counted drop.
__device__ __forceinline__ bool k1_resolve(const Dev& d, const uint4 a, const uint4 b, K1Local& L, K1Ev& e) {
    const u32 saddr = a.x, daddr = a.y;
    const u32 status = a.w & 0xFFFFu, proto = (a.w >> 16) & 0xFFu, flags = a.w >> 24;
    const bool alive = (flags & SG_EV_ALIVE) != 0;                // an open connection, not a request (data.go:1628-1679)
    const u32 label = alive ? 0u : a.z;                          // ... joined without a Host header
    e.dur = (u64)b.x | ((u64)b.y << 32); e.wt = (u64)b.z | ((u64)b.w << 32);
    e.dur = e.dur > SG_DUR_MAX ? SG_DUR_MAX : e.dur;
    e.alive = alive ? 1u : 0u;

    u32 spod = SG_NONE, ssvc = SG_NONE;
    join_pod_svc(d, saddr, spod, ssvc);
    if (spod == SG_NONE) { if (!alive) L.dsrc++; return false; }   // data.go:829-832: source must be a pod (:1643-1647 ignores silently)
    u32 from = SG_MAKE_REF(SG_REF_KNOWN, spod);
    const bool sharded = d.world > 1;
    u32 from_owner = sharded ? owner_hash_ref(from) : 0u;

    u32 dpod = SG_NONE, dsvc = SG_NONE, to, to_owner;
    join_pod_svc(d, daddr, dpod, dsvc);
    if (dsvc != SG_NONE) { to = SG_MAKE_REF(SG_REF_KNOWN, dsvc); to_owner = sharded ? owner_hash_ref(to) : 0u; }       // service first (:840-843)
    else if (dpod != SG_NONE) { to = SG_MAKE_REF(SG_REF_KNOWN, dpod); to_owner = sharded ? owner_hash_ref(to) : 0u; }  // then pod (:845-849)
    else if (label != 0) {                                       // outbound, Host header (:851-854)
        if (label > d.max_labels) { L.dcap++; return false; }
        to = SG_MAKE_REF(SG_REF_LABEL, label - 1); to_owner = owner_hash_ref(to);
        L.maxlabel = label > L.maxlabel ? label : L.maxlabel;
    } else {                                                     // outbound, raw IP (:862-863)
        u32 os;
        if (!table_slot(d.obkeys, d.obmask, (u64)daddr | (1ull << 32), 0ull, sg_fmix32(daddr) & d.obmask, os)) { L.dcap++; return false; }
        to = SG_MAKE_REF(SG_REF_OBIP, os); to_owner = owner_hash_obip(daddr);
    }
    if ((flags & SG_EV_REVERSE) && !alive) { u32 t = from; from = to; to = t; t = from_owner; from_owner = to_owner; to_owner = t; }  // dto.go:226-231
    if (sharded && (from_owner % d.world) != d.rank) { L.misr++; return false; }
    e.key = ((u64)from << 32) | (u64)to;
    if (alive) {                                                 // no request is counted; the key goes on the window's alive list
        e.err = 0; e.dur = 0;
        const u64 idx = atomicAdd(&d.ctr[C_ALIVE_N], 1ull);
        if (idx < d.alive_cap) d.alive_keys[idx] = e.key;
        return true;
    }
    e.err = is_error(proto, status);
    L.acc++;
    L.tmin = e.wt < L.tmin ? e.wt : L.tmin;
    L.tmax = e.wt > L.tmax ? e.wt : L.tmax;
    return true;
}

// per-workgroup statistics: wave reduce, then one update of this workgroup's private 64-byte line.
__device__ __forceinline__ void k1_publish_stats(const Dev& d, const K1Local& L) {
    const u64 tmin = wave_min_u64(L.tmin), tmax = wave_max_u64(L.tmax);
    const u32 ml = (u32)wave_max_u64(L.maxlabel);
    const u32 ds = wave_sum_u32(L.dsrc), dc = wave_sum_u32(L.dcap), mr = wave_sum_u32(L.misr), ac = wave_sum_u32(L.acc);
    if ((threadIdx.x & 63) == 0) {
        u64* w = d.wgstat + (size_t)(blockIdx.x % SG_MAX_K1_WGS) * WS_WORDS;
        if (ac) { atomicMin(&w[WS_TMIN], tmin); atomicMax(&w[WS_TMAX], tmax); atomicAdd(&w[WS_ACCEPTED], (u64)ac); }
        if (ml) atomicMax(&w[WS_MAXLABEL], (u64)ml);
        if (ds) atomicAdd(&w[WS_DROPPED_SRC], (u64)ds);
        if (dc) atomicAdd(&w[WS_DROPPED_CAP], (u64)dc);
        if (mr) atomicAdd(&w[WS_MISROUTED], (u64)mr);
    }
}

// join-table maintenance: the host's changed words, applied in stream order (one pair per word)
__global__ __launch_bounds__(256) void k_join_apply(u32* blob, const uint2* __restrict__ upd, u32 n) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) blob[upd[i].x] = upd[i].y;
}

// ---- variant 1: global open-addressing edge table + device-scope atomics.  General (any number
// of edges, any degree) but bound by the chip's ~22 G atomics/s and 12 ns per same-sector atomic
// (profiles/r01_atomic_probe.txt): kept for graphs the partitioned path cannot hold. ----------------
__global__ __launch_bounds__(256) void k1_resolve_aggregate(Dev d, const sg_event* __restrict__ ev, u64 n) {
    K1Local L; L.tmin = ~0ull; L.tmax = 0; L.maxlabel = L.dsrc = L.dcap = L.misr = L.acc = L.lost = 0;
    const uint4* __restrict__ p = reinterpret_cast<const uint4*>(ev);
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint4 a = p[2 * i], b = p[2 * i + 1];
        K1Ev e;
        if (!k1_resolve(d, a, b, L, e)) continue;
        u32 slot;
        if (!table_slot(d.ekeys, d.emask, e.key, SG_EKEY_EMPTY, hash_key64(e.key) & d.emask, slot)) { if (!e.alive) { L.dcap++; L.acc--; } continue; }
        if (e.alive) continue;                                       // the slot exists now (count 0): that is all an open connection adds here
        u64* acc = d.eacc + (size_t)slot * 4;
        const u64 us = e.dur / 1000ull;
        atomicAdd(&acc[0], 1ull | ((u64)e.err << 32));
        atomicAdd(&acc[1], e.dur);
        atomicMax(&acc[2], e.dur);
        atomicAdd(&acc[3], us * us);
        if (d.hist) atomicAdd(&d.hist_src[(size_t)slot * SG_HIST_BINS + hist_bin64(e.dur)], 1u);
    }
    k1_publish_stats(d, L);
}

// ---- variant 0: partitioned aggregation, no device-scope atomics on the event path. --------------
// Pass A (k1a_partition, per batch): one fat workgroup per CU streams a contiguous share of the batch.  Every event is
// joined against the LDS copy of the block table, its edge key hashed, and looked up in a first-come LDS cache; a cached
// key folds in with LDS atomics (hot edges collapse to one 40-byte aggregate per workgroup, which also keeps the
// partitions balanced: the hottest edge of C3 alone carries 1.3 % of the events), everything else leaves as a 16-byte
// single record for partition hash(key) / 2^k, into the piece (partition, this workgroup) — private to the workgroup, so
// the position inside it is an LDS counter.
// Pass B (k1b_merge, at window close): one workgroup per partition merges its pieces in an LDS table and writes each
// distinct edge once with plain stores.
#define K1A_THREADS 1024
#define K1A_G       4         // events per thread per step
#define K1A_NJ      6         // 16-byte join-blob words a thread stages into LDS (6 * 1024 * 16 B = 96 KiB at most)

// Issue a global load NOW and leave it in flight; a later s_waitcnt (inline asm that names the
// destination registers as in/out operands) is the matching wait.  Written as inline asm because the
// compiler puts waits between conditional loads.  vmcnt is in-order for loads, so the compiler's own
// (unaware) waits can only become stronger, never too weak.  Rule: no loop-carried value and no branch
// merge between an issue and its wait (a compiler-inserted register copy there would read a register that
// is still being loaded).
typedef u32 v4u_t __attribute__((ext_vector_type(4)));
typedef u32 v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gload16_issue(v4u_t& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory"); }

// Edge-key hash of both passes: 24-bit multiplies (full rate; a murmur finaliser is 4 quarter-rate 32-bit multiplies per
// key).  Node refs are small integers plus two type bits at the top: the low 24 bits go through the multipliers, the two
// top bytes through a third.  The partition is taken from the top bits (they depend on every input bit), cache bucket
// and pass-B table slot from lower bit ranges.  Balance on the C3 graph (1 M edges, 1024 partitions): sigma 29.7 edges
// against 31.3 for a Poisson split — indistinguishable from the finaliser.
__device__ __forceinline__ u32 edge_hash(u32 from, u32 to) {
    u32 x = __umul24(from, 0x9E3779u) + __umul24(to, 0x85EBCBu);
    x += __umul24((from >> 24) | ((to >> 24) << 8), 0xC2B2AFu);
    return __umul24(x >> 8, 0x27D4EBu);
}
__device__ __forceinline__ u32 part_of_hash(const Dev& d, u32 hk) { return hk >> (32u - (u32)__builtin_ctz(d.np)); }
__device__ __forceinline__ u32 part_of(const Dev& d, u64 key) { return part_of_hash(d, edge_hash((u32)(key >> 32), (u32)key)); }

// hb: 16 x u16 bins (8 words) of an aggregate, or nullptr (a single record's bin follows from its duration = a1)
__device__ __forceinline__ void ovf_append(const Dev& d, u32 p, u64 key, u64 a0, u64 a1, u64 a2, u64 a3, K1Local& L, const u32* hb = nullptr) {
    const u64 idx = atomicAdd(&d.ctr[C_OVF_N], 1ull);
    if (idx < d.ovf_cap) {
        u64* o = d.ovf + idx * 9; o[0] = key; o[1] = a0; o[2] = a1; o[3] = a2; o[4] = a3; d.ovf_p[idx] = p;
        if (d.hist) {                                                // words 5..8: the record's bins, 16 x u16
            u32* hw = reinterpret_cast<u32*>(o + 5);
            if (hb) { for (int j = 0; j < 8; j++) hw[j] = hb[j]; }
            else { for (int j = 0; j < 8; j++) hw[j] = 0; if (a0 & 0xFFFFFFFFull) { const u32 b = hist_bin64(a1); hw[b >> 1] = 1u << ((b & 1u) * 16); } }
        }
    }
    else { const u32 c = (u32)(a0 & 0xFFFFFFFFull); L.dcap += c; L.lost += c; }     // (the aggregate may carry other lanes' events: not L.acc -= c)
}
// exact for every 32-bit duration: floor(x / 1000) = (x * 0x10624DD3) >> 38
__device__ __forceinline__ u32 div1000_u32(u32 x) { return __umulhi(x, 0x10624DD3u) >> 6; }

// piece (p, w): pslots 16-byte slots; fc[p] = n_single | n_aggregate << 20 is this workgroup's LDS counter for it
__device__ __forceinline__ uint4* piece_of(const Dev& d, u32 p, u32 w) { return d.slab_s + ((size_t)p * d.nwg + w) * d.pslots; }
#define K1_NS(x) ((x) & 0xFFFFFu)
#define K1_NA(x) ((x) >> 20)
// zero = 1: a record that only creates the edge (SG_EV_ALIVE): count 0, all accumulators 0
__device__ __forceinline__ void emit_single(const Dev& d, u32* fc, u32 w, u32 p, u64 key, u64 dur, u32 err, K1Local& L, u32 zero = 0) {
    const u32 pos = K1_NS(atomicAdd(&fc[p], 1u));
    if (pos < d.ss) { if (!SG_ABL(d, 0x1u)) piece_of(d, SG_ABL(d, 0x40u) ? (p & 63u) : p, w)[pos] = make_uint4((u32)key, (u32)(key >> 32), (u32)dur, (u32)(dur >> 32) | (err << 31) | (zero << 30)); }
    else {
        atomicSub(&fc[p], 1u);                                       // the count stays exact (and below 2^20)
        if (zero) ovf_append(d, p, key, 0ull, 0ull, 0ull, 0ull, L);
        else { const u64 us = dur / 1000ull; ovf_append(d, p, key, 1ull | ((u64)err << 32), dur, dur, us * us, L); }
    }
}
// One aggregate of this launch's cache for piece (p, w).  A hot key produces one per launch and workgroup — always for the
// same piece — so in a window fed by many small batches the aggregates of EARLIER launches are searched first (they are the
// entries below the count the header held when this launch began; the piece is private to this workgroup and a key is flushed
// by exactly one lane, so the read-modify-write needs no atomics) and a match is updated in place.
// hb: the launch's 16 x u16 bins of the key (8 words, two bins each), or nullptr without the histogram
__device__ __forceinline__ void emit_agg(const Dev& d, u32* fc, u32 w, u32 p, u64 key, u64 a0, u64 a1, u64 a2, u64 a3, K1Local& L, bool first, const u32* hb) {
    uint4* ag = piece_of(d, p, w) + d.ss;
    const u32 AS = d.agg_slots;
    if (!first) {
        u32 na0 = K1_NA(d.hdr[(size_t)p * d.nwg + w]); na0 = na0 < d.sa ? na0 : d.sa;
        for (u32 r = 0; r < na0; r++) {
            uint4* o = ag + AS * r;
            const uint4 y0 = o[0];
            if (y0.x != (u32)key || y0.y != (u32)(key >> 32)) continue;
            u32 nb[8];
            if (hb) {                                                // the bins are 16-bit: merge only if none of them overflows
                const uint4 h0 = o[3], h1 = o[4];
                const u32 ob[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                bool fits = true;
#pragma unroll
                for (int j = 0; j < 8; j++) { const u32 lo = (ob[j] & 0xFFFFu) + (hb[j] & 0xFFFFu), hi = (ob[j] >> 16) + (hb[j] >> 16); fits &= lo <= 0xFFFFu && hi <= 0xFFFFu; nb[j] = lo | (hi << 16); }
                if (!fits) break;                                    // -> a second aggregate of the same key: pass B adds them up
            }
            const uint4 y1 = o[1]; const uint2 y2 = reinterpret_cast<const uint2*>(o + 2)[0];
            const u64 b0 = ((u64)y0.z | ((u64)y0.w << 32)) + a0, b1 = ((u64)y1.x | ((u64)y1.y << 32)) + a1;
            u64 b2 = (u64)y1.z | ((u64)y1.w << 32); b2 = a2 > b2 ? a2 : b2;
            const u64 b3 = ((u64)y2.x | ((u64)y2.y << 32)) + a3;
            o[0] = make_uint4(y0.x, y0.y, (u32)b0, (u32)(b0 >> 32));
            o[1] = make_uint4((u32)b1, (u32)(b1 >> 32), (u32)b2, (u32)(b2 >> 32));
            reinterpret_cast<uint2*>(o + 2)[0] = make_uint2((u32)b3, (u32)(b3 >> 32));
            if (hb) { o[3] = make_uint4(nb[0], nb[1], nb[2], nb[3]); o[4] = make_uint4(nb[4], nb[5], nb[6], nb[7]); }
            return;
        }
    }
    const u32 pos = K1_NA(atomicAdd(&fc[p], 1u << 20));
    if (pos < d.sa) {
        uint4* o = ag + AS * pos;
        o[0] = make_uint4((u32)key, (u32)(key >> 32), (u32)a0, (u32)(a0 >> 32));
        o[1] = make_uint4((u32)a1, (u32)(a1 >> 32), (u32)a2, (u32)(a2 >> 32));
        reinterpret_cast<uint2*>(o + 2)[0] = make_uint2((u32)a3, (u32)(a3 >> 32));
        if (hb) { o[3] = make_uint4(hb[0], hb[1], hb[2], hb[3]); o[4] = make_uint4(hb[4], hb[5], hb[6], hb[7]); }
    } else { atomicSub(&fc[p], 1u << 20); ovf_append(d, p, key, a0, a1, a2, a3, L, hb); }
}

// LDS edge cache of pass A: CT slots, a bucket = two adjacent key slots (one ds_read_b128 sees both).  The first two
// keys to arrive at a bucket own it for the launch (a slot never changes once it holds a key; every lane tries slot 0
// before slot 1, so a key cannot end up in both).  k0, k1: what the caller read from the bucket.  Returns the slot of
// `key`, or -1 (bucket owned by other keys).
__device__ __forceinline__ int cache_claim(u64* ckey, u32 bucket, u64 key, u64 k0, u64 k1) {
    if (k0 == SG_EKEY_EMPTY) { k0 = atomicCAS(&ckey[2 * bucket], SG_EKEY_EMPTY, key); if (k0 == SG_EKEY_EMPTY) k0 = key; }
    if (k0 == key) return (int)(2 * bucket);
    if (k1 == SG_EKEY_EMPTY) { k1 = atomicCAS(&ckey[2 * bucket + 1], SG_EKEY_EMPTY, key); if (k1 == SG_EKEY_EMPTY) k1 = key; }
    return k1 == key ? (int)(2 * bucket + 1) : -1;
}

template <bool L2LDS, bool SHARDED, bool HIST>
__global__ __launch_bounds__(K1A_THREADS) void k1a_partition(Dev d, const sg_event* __restrict__ ev, u64 n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 CT = d.k1a_ct;
    u64* ckey = reinterpret_cast<u64*>(smem);                       // [CT]
    u64* cacc = ckey + CT;                                           // [CT][4]
    u32* chist = reinterpret_cast<u32*>(cacc + (size_t)CT * 4);      // HIST: [CT][8] 16 x u16 bins per slot (a workgroup sees < 65536 events per launch)
    u32* fc = chist + (HIST ? (size_t)CT * 8 : 0);                   // [np]  n_single | n_aggregate << 20
    u64* red = reinterpret_cast<u64*>(fc + d.np);                    // [8] workgroup statistics (WS_* order)
    uint4* jl = reinterpret_cast<uint4*>(red + 8);                   // LDS copy of the join blob: jl1 | jl2 (L2LDS) | residual cuckoo (ck_in_lds)
    const u64* l1 = reinterpret_cast<const u64*>(jl);
    const u32* l2 = L2LDS ? reinterpret_cast<const u32*>(l1 + d.jl1mask + 1) : d.jl2;
    const u32 w = blockIdx.x, t = threadIdx.x;
    const uint4* __restrict__ pe = reinterpret_cast<const uint4*>(ev);
    const u64 per = (n + d.nwg - 1) / d.nwg;
    const u64 beg = (u64)w * per, end = (beg + per < n) ? beg + per : n;
    const bool first = d.batch_state == 1u;                          // first batch of the window: the headers are zero by definition
    if (beg >= end) {                                                // no share of this batch: pieces and statistics stay as they are,
        if (first) for (u32 p = t; p < d.np; p += K1A_THREADS) d.hdr[(size_t)p * d.nwg + w] = 0u;   // but stale headers must go
        return;
    }
    const u64 last = end - 1;
    SG_STAMP(d, 0, 0);
    static_assert(K1A_G == 4, "the event loads and the fold are written out for 4 events per lane");
    u64 i = beg + t;
    // K1A_G events per thread are fetched together (8 x 16 B in flight per lane); out-of-range lanes re-read the
    // share's last event and ignore it, so there is no branch between the loads.
#define K1A_ISSUE(base)                                                                                           \
        { const u64 j0 = (base), j1 = j0 + K1A_THREADS, j2 = j1 + K1A_THREADS, j3 = j2 + K1A_THREADS;               \
          const uint4* q0 = pe + 2 * (j0 < end ? j0 : last); const uint4* q1 = pe + 2 * (j1 < end ? j1 : last);     \
          const uint4* q2 = pe + 2 * (j2 < end ? j2 : last); const uint4* q3 = pe + 2 * (j3 < end ? j3 : last);     \
          gload16_issue(ea0, q0); gload16_issue(eb0, q0 + 1); gload16_issue(ea1, q1); gload16_issue(eb1, q1 + 1);   \
          gload16_issue(ea2, q2); gload16_issue(eb2, q2 + 1); gload16_issue(ea3, q3); gload16_issue(eb3, q3 + 1); }
    K1Local L; L.tmin = ~0ull; L.tmax = 0; L.maxlabel = L.dsrc = L.dcap = L.misr = L.acc = L.lost = 0;
    const u32 pshift = 32u - (u32)__builtin_ctz(d.np), bmask = CT / 2 - 1;
    const bool ck_any = d.ck_n != 0;

    // The general path (rare events: open connections, raw-IP outbound destinations, IPs in both maps or in the residual
    // cuckoo table, durations of 2^32 ns and more, labels out of range): the full join on the global tables.
    auto general = [&](const v4u_t va, const v4u_t vb) {
        K1Ev e;
        if (!k1_resolve(d, make_uint4(va.x, va.y, va.z, va.w), make_uint4(vb.x, vb.y, vb.z, vb.w), L, e)) return;
        const u32 hk = edge_hash((u32)(e.key >> 32), (u32)e.key), p = hk >> pshift;
        if (e.alive) { emit_single(d, fc, w, p, e.key, 0ull, 0u, L, 1u); return; }
        const u32 bkt = (hk >> 5) & bmask;
        const int slot = cache_claim(ckey, bkt, e.key, lds_fresh_u64(&ckey[2 * bkt]), lds_fresh_u64(&ckey[2 * bkt + 1]));
        if (slot >= 0) {
            const u64 us = e.dur / 1000ull;
            atomicAdd(&cacc[slot * 4], 1ull | ((u64)e.err << 32)); atomicAdd(&cacc[slot * 4 + 1], e.dur);
            atomicMax(&cacc[slot * 4 + 2], e.dur); atomicAdd(&cacc[slot * 4 + 3], us * us);
            if (HIST) { const u32 b = hist_bin64(e.dur); atomicAdd(&chist[slot * 8 + (b >> 1)], 1u << ((b & 1u) * 16)); }
        } else emit_single(d, fc, w, p, e.key, e.dur, e.err, L);
    };
    // The fast path, one event: branch-free join (two-level block table in LDS), data.go:827-870 as selects, cheap hash,
    // read-only cache probe.  `rare` hands the event to the general path instead.
    auto join = [&](u32 ip) -> u32 {
        const u32 b = ip >> 8;
        const u64 e1 = l1[((__umul24(b, SG_JL1_K1)) >> 9) & d.jl1mask], e2 = l1[((__umul24(b, SG_JL1_K2)) >> 11) & d.jl1mask];
        const u32 blk = (u32)e1 == b ? (u32)(e1 >> 32) : ((u32)e2 == b ? (u32)(e2 >> 32) : 0u);     // block 0 = the all-zero block
        return l2[(blk << 8) | (ip & 255u)];
    };
#define K1A_FAST(idx, va, vb, rare_out)                                                                             \
        {   const bool inr = (idx) < end;                                                                           \
            const u32 flags = (va).w >> 24, label = (va).z;                                                         \
            const u32 vs = join((va).x), vd = join((va).y);                                                         \
            const u32 ks = vs >> 30, kd = vd >> 30;                                                                 \
            bool rare = ((flags & SG_EV_ALIVE) != 0) | (ks == 3u) | (kd == 3u) | ((vb).y != 0u) |                   \
                        ((kd == 0u) & ((label == 0u) | (label > d.max_labels))) | (ck_any & ((vs == 0u) | (vd == 0u))); \
            rare &= inr; (rare_out) = rare;                                                                         \
            const bool fastv = inr & !rare;                                                                         \
            bool acc = fastv & (ks == 1u);                           /* data.go:829-832: the source must be a pod */ \
            L.dsrc += (fastv & (ks != 1u)) ? 1u : 0u;                                                               \
            u32 from = vs & 0x3FFFFFFFu;                                                                            \
            u32 to = kd ? (vd & 0x3FFFFFFFu) : (SG_MAKE_REF(SG_REF_LABEL, label - 1u));   /* service / pod id, else Host label (:840-854) */ \
            { const u32 ml = (acc & (kd == 0u)) ? label : 0u; L.maxlabel = ml > L.maxlabel ? ml : L.maxlabel; }     \
            if (flags & SG_EV_REVERSE) { const u32 x_ = from; from = to; to = x_; }      /* dto.go:226-231 */       \
            if (SHARDED) { const bool mine = (owner_hash_ref(from) % d.world) == d.rank; L.misr += (acc & !mine) ? 1u : 0u; acc &= mine; } \
            const u32 status = (va).w & 0xFFFFu, proto = ((va).w >> 16) & 0xFFu, dur = (vb).x;                      \
            const u32 err = is_error(proto, status);                                                                \
            const u64 wt = (u64)(vb).z | ((u64)(vb).w << 32);                                                       \
            L.acc += acc ? 1u : 0u;                                                                                 \
            L.tmin = (acc && wt < L.tmin) ? wt : L.tmin; L.tmax = (acc && wt > L.tmax) ? wt : L.tmax;              \
            const u32 hk = edge_hash(from, to), part = hk >> pshift, bucket = (hk >> 5) & bmask;                    \
            const u64 key = ((u64)from << 32) | (u64)to;                                                            \
            const ulonglong2 kk = reinterpret_cast<const ulonglong2*>(ckey)[bucket];                                \
            int slot = kk.x == key ? (int)(2u * bucket) : (kk.y == key ? (int)(2u * bucket + 1u) : -1);             \
            if (acc && slot < 0 && (kk.x == SG_EKEY_EMPTY || kk.y == SG_EKEY_EMPTY)) slot = cache_claim(ckey, bucket, key, kk.x, kk.y); \
            if SG_ABL(d, 0x2u) slot = -1;                                                                         \
            if (acc && !SG_ABL(d, 0x8u)) {                                                                        \
                if (slot >= 0 && !SG_ABL(d, 0x4u)) {                                                              \
                    const u32 us = div1000_u32(dur);                                                                \
                    const u64 ssq = (u64)us * (u64)us;                        /* us < 2^23: 24-bit multiplies */             \
                    atomicAdd(&cacc[slot * 4], 1ull | ((u64)err << 32)); atomicAdd(&cacc[slot * 4 + 1], (u64)dur);  \
                    atomicMax(&cacc[slot * 4 + 2], (u64)dur); atomicAdd(&cacc[slot * 4 + 3], ssq);                  \
                    if (HIST) { const u32 hb_ = hist_bin32(dur); atomicAdd(&chist[slot * 8 + (hb_ >> 1)], 1u << ((hb_ & 1u) * 16)); } \
                } else emit_single(d, fc, w, part, key, (u64)dur, err, L);                                          \
            }                                                                                                       \
        }
#define K1A_FOLD(base)                                                                                            \
        { asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea0), "+v"(eb0), "+v"(ea1), "+v"(eb1), "+v"(ea2), "+v"(eb2), "+v"(ea3), "+v"(eb3) : : "memory"); \
          bool r0, r1, r2, r3;                                                                                      \
          K1A_FAST((base), ea0, eb0, r0) K1A_FAST((base) + K1A_THREADS, ea1, eb1, r1)                               \
          K1A_FAST((base) + 2 * K1A_THREADS, ea2, eb2, r2) K1A_FAST((base) + 3 * K1A_THREADS, ea3, eb3, r3)         \
          if (__builtin_amdgcn_ballot_w64(r0 | r1 | r2 | r3)) {             /* one copy of the general path: register selects */ \
              _Pragma("unroll 1")                                                                                   \
              for (u32 q = 0; q < K1A_G; q++) {                                                                     \
                  const bool rq = q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : r3;                                     \
                  if (!__builtin_amdgcn_ballot_w64(rq)) continue;                                                   \
                  const v4u_t va = q == 0 ? ea0 : q == 1 ? ea1 : q == 2 ? ea2 : ea3;                                \
                  const v4u_t vb = q == 0 ? eb0 : q == 1 ? eb1 : q == 2 ? eb2 : eb3;                                \
                  if (rq) general(va, vb);                                                                          \
              } } }
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory")   /* LDS-only: does not drain the global stores */
    {
        // piece counters: zero by definition in the first batch of a window (no loads); a later batch reads them with
        // ordinary loads BEFORE anything is issued by hand
        for (u32 p = t; p < d.np; p += K1A_THREADS) fc[p] = first ? 0u : d.hdr[(size_t)p * d.nwg + w];
        // the LDS set-up happens BEFORE anything is issued by hand: between a hand-issued load and its wait there must be no
        // code at all (a loop there once made the register allocator move in-flight registers: the staged join table came
        // out as garbage -> wild record addresses -> a memory fault, or a few hundred silently lost events)
        for (u32 k = t; k < CT; k += K1A_THREADS) ckey[k] = SG_EKEY_EMPTY;
        for (u32 k = t; k < CT * 4; k += K1A_THREADS) cacc[k] = 0;
        if (HIST) for (u32 k = t; k < CT * 8; k += K1A_THREADS) chist[k] = 0;
        if (t < 8) red[t] = t == WS_TMIN ? ~0ull : 0ull;
        // The join blob goes out first; right behind it one load per event of the first group, into a register nobody
        // reads: it pulls the group's lines towards this XCD's L2 while the LDS is being set up (every hand-issued load is
        // waited for inside the straight-line region that issued it, so the first group's real loads belong to the loop).
        v4u_t jb0, jb1, jb2, jb3, jb4, jb5; u32 pf;
        static_assert(K1A_NJ == 6, "written out for 6 blob words per lane");
        const u32 n16 = d.jstage_bytes >> 4, n1 = (d.jl1mask + 1) >> 1;   // 16-byte words to stage; of them level 1 (always there)
        const uint4* g1 = reinterpret_cast<const uint4*>(d.jl1); const uint4* g2 = reinterpret_cast<const uint4*>(d.jl2) - n1;
#define K1A_JIDX(k) ((t + (k) * K1A_THREADS) < n16 ? (t + (k) * K1A_THREADS) : n16 - 1)
#define K1A_JSRC(k) ((K1A_JIDX(k) < n1 ? g1 : g2) + K1A_JIDX(k))
        const uint4* js0 = K1A_JSRC(0); const uint4* js1 = K1A_JSRC(1); const uint4* js2 = K1A_JSRC(2);
        const uint4* js3 = K1A_JSRC(3); const uint4* js4 = K1A_JSRC(4); const uint4* js5 = K1A_JSRC(5);
        const uint4* pf0 = pe + 2 * (i < end ? i : last); const uint4* pf1 = pe + 2 * (i + K1A_THREADS < end ? i + K1A_THREADS : last);
        const uint4* pf2 = pe + 2 * (i + 2 * K1A_THREADS < end ? i + 2 * K1A_THREADS : last); const uint4* pf3 = pe + 2 * (i + 3 * K1A_THREADS < end ? i + 3 * K1A_THREADS : last);
        // all addresses are in registers: ten issues and the wait in ONE statement
        asm volatile("global_load_dwordx4 %0, %7, off\n\tglobal_load_dwordx4 %1, %8, off\n\tglobal_load_dwordx4 %2, %9, off\n\t"
                     "global_load_dwordx4 %3, %10, off\n\tglobal_load_dwordx4 %4, %11, off\n\tglobal_load_dwordx4 %5, %12, off\n\t"
                     "global_load_dword %6, %13, off\n\tglobal_load_dword %6, %14, off\n\tglobal_load_dword %6, %15, off\n\tglobal_load_dword %6, %16, off\n\t"
                     "s_waitcnt vmcnt(4)"
                     : "=&v"(jb0), "=&v"(jb1), "=&v"(jb2), "=&v"(jb3), "=&v"(jb4), "=&v"(jb5), "=&v"(pf)
                     : "v"(js0), "v"(js1), "v"(js2), "v"(js3), "v"(js4), "v"(js5), "v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3) : "memory");
#define K1A_JST(k, r) if (t + (k) * K1A_THREADS < n16) jl[t + (k) * K1A_THREADS] = make_uint4((r).x, (r).y, (r).z, (r).w)
        K1A_JST(0, jb0); K1A_JST(1, jb1); K1A_JST(2, jb2); K1A_JST(3, jb3); K1A_JST(4, jb4); K1A_JST(5, jb5);
#undef K1A_JST
#undef K1A_JSRC
#undef K1A_JIDX
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf) : : "memory");
        LDS_BARRIER();
        SG_STAMP(d, 0, 1);
    }
    for (; i < end; i += (u64)K1A_G * K1A_THREADS) {
        v4u_t ea0, eb0, ea1, eb1, ea2, eb2, ea3, eb3;
        K1A_ISSUE(i);
        K1A_FOLD(i);
    }
    SG_STAMP(d, 0, 3);
#undef K1A_ISSUE
#undef K1A_FOLD
#undef K1A_FAST
    LDS_BARRIER();
    SG_STAMP(d, 0, 4);
    // flush the cache, singles first (they share the singles region with the loop's records), then the aggregates
    for (u32 s = t; s < CT; s += K1A_THREADS) {
        const u64 k = ckey[s];
        if (k == SG_EKEY_EMPTY) continue;
        const u64 x0 = cacc[s * 4];
        if ((x0 & 0xFFFFFFFFull) == 1ull) emit_single(d, fc, w, part_of(d, k), k, cacc[s * 4 + 1], (u32)(x0 >> 32), L);
        else if ((x0 & 0xFFFFFFFFull) != 0ull) {
            u32 hb[8];
            if (HIST) { for (int j = 0; j < 8; j++) hb[j] = chist[s * 8 + j]; }
            emit_agg(d, fc, w, part_of(d, k), k, x0, cacc[s * 4 + 1], cacc[s * 4 + 2], cacc[s * 4 + 3], L, first, HIST ? hb : nullptr);
        }
    }
    // workgroup statistics: wave reduce -> LDS -> one thread updates this workgroup's private line
    {
        const u64 tmin = wave_min_u64(L.tmin), tmax = wave_max_u64(L.tmax);
        const u32 ml = (u32)wave_max_u64(L.maxlabel);
        const u32 ds = wave_sum_u32(L.dsrc), dc = wave_sum_u32(L.dcap), mr = wave_sum_u32(L.misr), ac = wave_sum_u32(L.acc), ls = wave_sum_u32(L.lost);
        if ((t & 63) == 0) {
            if (ac) { atomicMin(&red[WS_TMIN], tmin); atomicMax(&red[WS_TMAX], tmax); atomicAdd(&red[WS_ACCEPTED], (u64)ac); }
            if (ls) atomicAdd(&red[WS_PAD], (u64)ls);
            if (ml) atomicMax(&red[WS_MAXLABEL], (u64)ml);
            if (ds) atomicAdd(&red[WS_DROPPED_SRC], (u64)ds);
            if (dc) atomicAdd(&red[WS_DROPPED_CAP], (u64)dc);
            if (mr) atomicAdd(&red[WS_MISROUTED], (u64)mr);
        }
    }
    LDS_BARRIER();
    for (u32 p = t; p < d.np; p += K1A_THREADS) d.hdr[(size_t)p * d.nwg + w] = fc[p];
    SG_STAMP(d, 0, 5);
    if (t == 0) {
        u64* g = d.wgstat + (size_t)(blockIdx.x % SG_MAX_K1_WGS) * WS_WORDS;
        // accepted = counted by the lanes - dropped afterwards for capacity (a workgroup only drops what it accepted itself)
        if (red[WS_ACCEPTED]) { atomicMin(&g[WS_TMIN], red[WS_TMIN]); atomicMax(&g[WS_TMAX], red[WS_TMAX]); atomicAdd(&g[WS_ACCEPTED], red[WS_ACCEPTED] - red[WS_PAD]); }
        if (red[WS_MAXLABEL]) atomicMax(&g[WS_MAXLABEL], red[WS_MAXLABEL]);
        if (red[WS_DROPPED_SRC]) atomicAdd(&g[WS_DROPPED_SRC], red[WS_DROPPED_SRC]);
        if (red[WS_DROPPED_CAP]) atomicAdd(&g[WS_DROPPED_CAP], red[WS_DROPPED_CAP]);
        if (red[WS_MISROUTED]) atomicAdd(&g[WS_MISROUTED], red[WS_MISROUTED]);
    }
    SG_STAMP(d, 0, 6);
#undef LDS_BARRIER
}

// Pass B.  Workgroup p owns partition p: it reads the record counts of its nwg pieces (one contiguous line of d.hdr),
// then exactly the records that exist (K1B_U single records per lane in flight, the lanes of a piece side by side),
// merges them in an LDS table and writes every distinct edge once with plain stores:
//   e_from/e_to [p*pcap + i]  dense endpoints        acc_src [(p*pcap + i)*4]  accumulators
//   deg[from][replica] += 1 (atomic u32; a row's edges are spread over the partitions; the returned value is the
//   edge's position inside its CSR row's replica)
// LDS: the table and nothing else — k1b_ht * 40 bytes; with 2048 slots that is exactly half of a CU's 160 KiB, so two
// 512-thread workgroups share a CU and one's (latency-bound) header round trip and compaction overlap the other's
// (LDS-atomic-bound) merge.  The last key slot is never used as a slot: its 8 bytes hold the two workgroup counters.
// NSG: scalar-register cap.  72 when a CU gets several partitions (above 80 SGPRs a CU holds ONE 1024-thread workgroup, below it
// two, tools/occupancy_probe.hip); uncapped (k1b_merge_wide: no scalar spills) when every CU has at most one partition anyway
// (C2: 15.4 vs 16.8 us).
template <int K1B_U, bool HIST>   // K1B_U: single records a lane has in flight; HIST: per-edge latency histogram (f-3)
__device__ __forceinline__ void k1b_body(const Dev& d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 HT = d.k1b_ht, hmask = HT - 1;
    u64* hkey = reinterpret_cast<u64*>(smem);                       // [HT]  (slot HT-1: n_drop, out_n)
    u64* hacc = hkey + HT;                                           // [4][HT]: accumulator j of slot h at j*HT + h — an 8-byte stride across
                                                                     // lanes (a 32-byte stride puts a lane group's 16 addresses on 4 bank pairs)
    u32* hh = reinterpret_cast<u32*>(hacc + (size_t)HT * 4);         // HIST: [HT][16] bins
    u32* n_drop = reinterpret_cast<u32*>(hkey + hmask); u32* out_n = n_drop + 1;
    const u32 p = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
    SG_STAMP(d, 1, 0);
    const bool empty = d.batch_state == 2u;                          // no batch this window: the pieces are the previous window's
    // LPP lanes walk one piece; each reads the piece's header word itself (a partition's headers are one contiguous KiB)
    const u32 LPP = NT > d.nwg ? NT / d.nwg : 1u;
    const u32 sub = t % LPP;
    const u32 w0 = t / LPP;
    u32 h0 = (!empty && w0 < d.nwg) ? d.hdr[(size_t)p * d.nwg + w0] : 0u;
    // the first K1B_U records of the lane's first piece go out together with the header word, not after it (index clamped to
    // the piece's capacity; what lies beyond the count is ignored): one round trip instead of two, hidden behind the table set-up
    uint4 xf[K1B_U];
    {
        const uint4* piece0 = piece_of(d, p, w0 < d.nwg ? w0 : 0u);
        const u32 ssm1 = d.ss - 1;
#pragma unroll
        for (int u = 0; u < K1B_U; u++) { const u32 r = sub + (u32)u * LPP; xf[u] = piece0[r < ssm1 ? r : ssm1]; }
    }
    // counters the tail needs: fetched now so their latency hides behind the merge
    const u64 ovf_n = d.ctr[C_OVF_N];
    const u32 nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS], nob = (u32)d.ctr[C_N_OBIP];
    for (u32 i = t; i < hmask; i += NT) hkey[i] = SG_EKEY_EMPTY;
    for (u32 i = t; i < HT * 4; i += NT) hacc[i] = 0;
    if (HIST) for (u32 i = t; i < HT * SG_HIST_BINS; i += NT) hh[i] = 0;
    if (t == 0) { *n_drop = 0; *out_n = 0; }
    __syncthreads();
    SG_STAMP(d, 1, 1);

    // bins: nullptr = a single record (its bin follows from the duration a1, if it counts a request), else 16 x u16 in 8 words
    auto add = [&](u64 key, u64 a0, u64 a1, u64 a2, u64 a3, const u32* bins) {
        u32 h = (edge_hash((u32)(key >> 32), (u32)key) >> 4) & hmask; bool ok = false;
        h = h == hmask ? 0u : h;
        for (u32 it = 0; it < HT; it++) {                            // bounded: the table holds at most HT - 1 distinct edges
            u64 k = lds_fresh_u64(&hkey[h]);
            if (k == SG_EKEY_EMPTY) { k = atomicCAS(&hkey[h], SG_EKEY_EMPTY, key); if (k == SG_EKEY_EMPTY) k = key; }
            if (k == key) { ok = true; break; }
            h = h + 1 >= hmask ? 0u : h + 1;
        }
        if (!ok) { atomicAdd(n_drop, (u32)(a0 & 0xFFFFFFFFull)); return; }
        atomicAdd(&hacc[h], a0); atomicAdd(&hacc[HT + h], a1); atomicMax(&hacc[2 * HT + h], a2); atomicAdd(&hacc[3 * HT + h], a3);
        if (HIST) {
            if (!bins) { if (a0 & 0xFFFFFFFFull) atomicAdd(&hh[h * SG_HIST_BINS + hist_bin64(a1)], 1u); }
            else for (u32 j = 0; j < 8; j++) {
                if (bins[j] & 0xFFFFu) atomicAdd(&hh[h * SG_HIST_BINS + 2 * j], bins[j] & 0xFFFFu);
                if (bins[j] >> 16) atomicAdd(&hh[h * SG_HIST_BINS + 2 * j + 1], bins[j] >> 16);
            }
        }
    };
    // record r of a piece belongs to lane r % LPP of its group; every load is unconditional (index clamped to the piece's
    // last record, result ignored) so that the K1B_U of a round are in flight together
    for (u32 w = w0; w < d.nwg; w += NT / LPP) {
        const u32 h = w == w0 ? h0 : d.hdr[(size_t)p * d.nwg + w];
        const u32 ns = K1_NS(h) < d.ss ? K1_NS(h) : d.ss, na = K1_NA(h) < d.sa ? K1_NA(h) : d.sa;
        if (!(ns | na)) continue;
        const uint4* piece = piece_of(d, p, w);
        const u32 lastr = ns ? ns - 1 : 0;
        for (u32 r0 = sub; r0 < ns; r0 += LPP * K1B_U) {
            uint4 x[K1B_U];
            if (w == w0 && r0 == sub) {
#pragma unroll
                for (int u = 0; u < K1B_U; u++) x[u] = xf[u];
            } else {
#pragma unroll
                for (int u = 0; u < K1B_U; u++) { const u32 r = r0 + u * LPP; x[u] = piece[r < ns ? r : lastr]; }
            }
#pragma unroll
            for (int u = 0; u < K1B_U; u++) if (r0 + u * LPP < ns) {
                const u64 key = (u64)x[u].x | ((u64)x[u].y << 32);
                const u32 dhi = x[u].w & 0x3FFFFFFFu;
                const u64 dur = (u64)x[u].z | ((u64)dhi << 32);
                u64 ssq;
                if (dhi == 0) { const u32 us = div1000_u32(x[u].z); ssq = (u64)us * (u64)us; }
                else { const u64 us = dur / 1000ull; ssq = us * us; }
                const u64 one = ((x[u].w >> 30) & 1u) ? 0ull : 1ull;             // bit 62: edge-only record (SG_EV_ALIVE)
                if (!SG_ABL(d, 0x10u)) add(key, one | ((u64)(x[u].w >> 31) << 32), dur, dur, ssq, nullptr);
            }
        }
        for (u32 r = sub; r < na; r += LPP) {
            const uint4* q = piece + d.ss + d.agg_slots * r;
            const uint4 y0 = q[0], y1 = q[1]; const uint2 y2 = reinterpret_cast<const uint2*>(q + 2)[0];
            u32 hb[8];
            if (HIST) { const uint4 h0 = q[3], h1 = q[4]; hb[0] = h0.x; hb[1] = h0.y; hb[2] = h0.z; hb[3] = h0.w; hb[4] = h1.x; hb[5] = h1.y; hb[6] = h1.z; hb[7] = h1.w; }
            add((u64)y0.x | ((u64)y0.y << 32), (u64)y0.z | ((u64)y0.w << 32), (u64)y1.x | ((u64)y1.y << 32),
                (u64)y1.z | ((u64)y1.w << 32), (u64)y2.x | ((u64)y2.y << 32), hb);     // (an aggregate always carries bins in HIST mode; hb is ignored otherwise)
        }
    }
    __syncthreads();
    SG_STAMP(d, 1, 3);
    // (no reset of the headers: the first batch of the next window rewrites every one of them)
    if (ovf_n) {
        const u64 no = ovf_n < d.ovf_cap ? ovf_n : d.ovf_cap;
        for (u64 i = t; i < no; i += NT) {
            if (d.ovf_p[i] != p) continue;
            const u64* o = d.ovf + i * 9;
            add(o[0], o[1], o[2], o[3], o[4], reinterpret_cast<const u32*>(o + 5));   // (overflow records always carry their bins as 16 x u16)
        }
        __syncthreads();
    }
    SG_STAMP(d, 1, 4);

    // compact the table into the partition's output slots (order within a partition is arbitrary;
    // the CSR row sort makes the final order canonical)
    // Two table slots per thread at most (k1b_ht <= 2 x threads): the returning `deg` atomics of both are issued before
    // either result is stored — one round trip per partition instead of two.
    for (u32 s0 = t; s0 < hmask; s0 += 2 * NT) {
        u32 f[2], to[2], oi[2], rk[2]; bool live[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            const u32 s = s0 + (u32)k2 * NT;
            live[k2] = false; f[k2] = to[k2] = oi[k2] = rk[k2] = 0;
            if (s >= hmask) continue;
            const u64 k = hkey[s];
            if (k == SG_EKEY_EMPTY) continue;
            f[k2] = dense_of(d, (u32)(k >> 32), nk, nl, nob); to[k2] = dense_of(d, (u32)k, nk, nl, nob);
            if (f[k2] == SG_NONE || to[k2] == SG_NONE) { atomicAdd(n_drop, (u32)(hacc[s] & 0xFFFFFFFFull)); continue; }
            oi[k2] = atomicAdd(out_n, 1u);
            if (oi[k2] >= d.pcap) { atomicAdd(n_drop, (u32)(hacc[s] & 0xFFFFFFFFull)); continue; }
            live[k2] = true;
            if (!d.dh_g) rk[k2] = atomicAdd(&d.deg[SG_DEG_IDX(f[k2], p & (SG_DEG_REP - 1))], 1u);     // arrival order inside the row's replica (dh_g: k2_deg_hist ranks the edges)
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            if (!live[k2]) continue;
            const u32 s = s0 + (u32)k2 * NT;
            const size_t slot = (size_t)p * d.pcap + oi[k2];
            d.e_from[slot] = f[k2]; d.e_to[slot] = to[k2];
            ulonglong2* o = reinterpret_cast<ulonglong2*>(d.acc_src + slot * 4);
            o[0] = make_ulonglong2(hacc[s], hacc[HT + s]); o[1] = make_ulonglong2(hacc[2 * HT + s], hacc[3 * HT + s]);
            if (HIST) {
                uint4* ho = reinterpret_cast<uint4*>(d.hist_src + slot * SG_HIST_BINS); const u32* hs = hh + s * SG_HIST_BINS;
                ho[0] = make_uint4(hs[0], hs[1], hs[2], hs[3]); ho[1] = make_uint4(hs[4], hs[5], hs[6], hs[7]);
                ho[2] = make_uint4(hs[8], hs[9], hs[10], hs[11]); ho[3] = make_uint4(hs[12], hs[13], hs[14], hs[15]);
            }
            if (!d.dh_g) d.e_rank[slot] = rk[k2];
        }
    }
    __syncthreads();
    SG_STAMP(d, 1, 5);
    if (t == 0) {
        const u32 on = *out_n, nd = *n_drop;
        d.part_n[p] = on < d.pcap ? on : d.pcap;
        if (nd) {                                                    // dropped after pass A had counted them as accepted
            atomicAdd(&d.ctr[C_DROPPED_CAP], (u64)nd);
            atomicAdd(&d.ctr[C_N_EVENTS], 0ull - (u64)nd);
        }
    }
}

template <int K1B_U, bool HIST> __global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(72))) void k1b_merge(Dev d) { k1b_body<K1B_U, HIST>(d); }
template <int K1B_U, bool HIST> __global__ __launch_bounds__(1024) void k1b_merge_wide(Dev d) { k1b_body<K1B_U, HIST>(d); }

#include "sg_k1_narrow.h"   // the narrow-record form of both passes (default of variant 0)
#include "sg_k1_team.h"     // round 4: pass A with two teams per workgroup and a batched join

// ------------------------------------------------------------------------------------------------
// K2  csr_build: canonical node numbering, CSR with sorted rows.
// ------------------------------------------------------------------------------------------------
// one workgroup (1024 threads): window bookkeeping.
//   (a) fold the per-workgroup K1 statistics into the counters and re-arm the slots;
//   (b) collect the window's distinct raw outbound IPs (or take the sharded driver's union list),
//       sort them (bitonic, global memory), drop duplicates -> ob_sorted, N_OBIP;
//   (c) N = NK + NL + NOB.
// Warm windows: thread 0 also decides whether this window may take the warm path at all (C_COLD = 0): the host wants to try
// (warm_try), the kept state is whole, the node numbering it was written in still holds (same N_KNOWN and N_LABELS — the dense
// ids of labels follow the known nodes'), and the window has no raw outbound IP (their dense ids are ranks among the window's own).
__device__ __forceinline__ void kc_warm_decide(const Dev& d, u32 warm_try, u64 n_known, u64 nl, u64 nob) {
    if (!d.warm) return;
    (void)n_known; (void)nl;                                         // (until round 5 the kept columns were dense ids: N_KNOWN and N_LABELS had to be what they were at capture)
    const bool ok = warm_try && d.ctr[C_KEPT_VALID] && d.ctr[C_KEPT_E] != 0 && nob == 0;
    d.ctr[C_COLD] = ok ? 0ull : 1ull;
}
__global__ __launch_bounds__(1024) void kc_prepare(Dev d, u64 n_known, u64 n_labels_decl, u32* list, const u32* n_in, u32 list_cap, u32 collect,
                                                   const u32* seg, u32 seg_stride, u32 seg_world, u32 warm_try) {
    __shared__ u64 red[7][16];
    __shared__ u32 wsum[17];
    __shared__ u32 cnt;
    __shared__ u64 nkl;                                              // N_KNOWN + N_LABELS as thread 0 wrote them (no second trip to memory for step (c))
    __shared__ u64 nl_s;
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // (the first stretch of the outbound-IP table travels together with the statistics: one round trip to memory, not two — this
    // kernel is one workgroup, the chip waits for it, and it is nothing but dependent round trips)
    const u64 ob0 = (collect == 1 && t <= d.obmask) ? d.obkeys[t] : 0ull;
    const u64 nl_prev = t == 0 ? d.ctr[C_N_LABELS] : 0ull;           // (so does the label count of the windows before: it was a third trip, behind the barrier)
    // (a)
    {
        u64 tmin = ~0ull, tmax = 0, ml = 0, ds = 0, dc = 0, mr = 0, ac = 0;
        const u32 nslots = d.variant == 0 ? d.nwg : SG_MAX_K1_WGS;
        for (u32 i = t; i < nslots; i += 1024) {
            u64* w = d.wgstat + (size_t)i * WS_WORDS;
            tmin = w[WS_TMIN] < tmin ? w[WS_TMIN] : tmin; tmax = w[WS_TMAX] > tmax ? w[WS_TMAX] : tmax;
            ml = w[WS_MAXLABEL] > ml ? w[WS_MAXLABEL] : ml;
            ds += w[WS_DROPPED_SRC]; dc += w[WS_DROPPED_CAP]; mr += w[WS_MISROUTED]; ac += w[WS_ACCEPTED];
            w[WS_TMIN] = ~0ull; w[WS_TMAX] = 0; w[WS_MAXLABEL] = 0; w[WS_DROPPED_SRC] = 0; w[WS_DROPPED_CAP] = 0; w[WS_MISROUTED] = 0; w[WS_ACCEPTED] = 0;
        }
        tmin = wave_min_u64(tmin); tmax = wave_max_u64(tmax); ml = wave_max_u64(ml);
        ds = wave_sum_u64(ds); dc = wave_sum_u64(dc); mr = wave_sum_u64(mr); ac = wave_sum_u64(ac);
        if (lane == 0) { red[0][wave] = tmin; red[1][wave] = tmax; red[2][wave] = ml; red[3][wave] = ds; red[4][wave] = dc; red[5][wave] = mr; red[6][wave] = ac; }
        if (t == 0) cnt = 0;
        __syncthreads();
        if (t == 0) {
            for (int k = 1; k < 16; k++) {
                red[0][0] = red[0][k] < red[0][0] ? red[0][k] : red[0][0]; red[1][0] = red[1][k] > red[1][0] ? red[1][k] : red[1][0];
                red[2][0] = red[2][k] > red[2][0] ? red[2][k] : red[2][0];
                red[3][0] += red[3][k]; red[4][0] += red[4][k]; red[5][0] += red[5][k]; red[6][0] += red[6][k];
            }
            d.ctr[C_TMIN_NS] = red[0][0]; d.ctr[C_TMAX_NS] = red[1][0];
            u64 nl = nl_prev;                                        // labels are cumulative across windows
            nl = red[2][0] > nl ? red[2][0] : nl; nl = n_labels_decl > nl ? n_labels_decl : nl;
            d.ctr[C_N_LABELS] = nl; d.ctr[C_N_KNOWN] = n_known; nkl = nl + n_known; nl_s = nl;
            d.ctr[C_DROPPED_SRC] = red[3][0]; d.ctr[C_MISROUTED] = red[5][0]; d.ctr[C_N_EVENTS] = red[6][0];
            d.ctr[C_DROPPED_CAP] = red[4][0];                        // K1b / K2 add their own drops afterwards
            d.ctr[C_DELTA_N] = 0;                                    // the warm pass B counts the window's new edges
            d.ctr[C_N_LONG] = 0;                                     // k2_rowptr's workgroups append to the long-row list
            d.ctr[C_HUB_ITEMS] = 0;                                  // ... and to the hub-block work list
        }
    }
    // (b)
    u32 n;
    if (collect == 1) {
        for (u32 i = t; i <= d.obmask; i += 1024) {
            const u64 k = i == t ? ob0 : d.obkeys[i];
            if (k) { const u32 pos = atomicAdd(&cnt, 1u); if (pos < list_cap) list[pos] = (u32)k; }
        }
        __syncthreads();
        n = cnt < list_cap ? cnt : list_cap;
    } else if (collect == 2) {
        // all-gathered per-shard lists: seg[r * seg_stride] = count, entries follow (sharded driver, no host sync)
        u32 off = 0;
        for (u32 r = 0; r < seg_world; r++) {
            const u32* sr = seg + (size_t)r * seg_stride;
            const u32 c = sr[0] < seg_stride - 1 ? sr[0] : seg_stride - 1;
            for (u32 i = t; i < c; i += 1024) if (off + i < list_cap) list[off + i] = sr[1 + i];
            off += c;
        }
        __syncthreads();
        n = off < list_cap ? off : list_cap;
    } else {
        n = *n_in < list_cap ? *n_in : list_cap;
    }
    if (n == 0) {                                                    // (uniform) no raw outbound IP this window: nothing to sort or to number
        if (t == 0) { d.ctr[C_N_OBIP] = 0; d.ctr[C_N_NODES] = nkl; kc_warm_decide(d, warm_try, n_known, nl_s, 0); }
        return;
    }
    u32 np2 = 1; while (np2 < n) np2 <<= 1;
    for (u32 i = n + t; i < np2; i += 1024) list[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (u32 k = 2; k <= np2; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = t; i < np2; i += 1024) {
                const u32 x = i ^ j;
                if (x > i) {
                    const u32 a = list[i], b = list[x];
                    if ((a > b) == ((i & k) == 0)) { list[i] = b; list[x] = a; }
                }
            }
            __syncthreads();
        }
    const u32 per = (n + 1023) / 1024;
    const u32 beg = t * per < n ? t * per : n, end = (beg + per < n) ? beg + per : n;
    u32 c = 0;
    for (u32 i = beg; i < end; i++) c += (i == 0 || list[i] != list[i - 1]) ? 1u : 0u;
    u32 total;
    u32 pos = block_excl_scan<1024>(c, wsum, &total);
    for (u32 i = beg; i < end; i++) if (i == 0 || list[i] != list[i - 1]) { if (pos < d.max_obip) d.ob_sorted[pos] = list[i]; pos++; }
    if (t == 0) {
        const u64 nob = total < d.max_obip ? total : d.max_obip;
        d.ctr[C_N_OBIP] = nob;
        d.ctr[C_N_NODES] = nkl + nob;
        kc_warm_decide(d, warm_try, n_known, nl_s, nob);
    }
}

// one workgroup: the window's distinct raw outbound IPs into a caller-owned list (sharded driver).
__global__ __launch_bounds__(1024) void k2_ob_collect(Dev d, u32* list, u32 list_cap, u32* n_out) {
    __shared__ u32 cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (u32 i = threadIdx.x; i <= d.obmask; i += 1024) {
        const u64 k = d.obkeys[i];
        if (k) { const u32 pos = atomicAdd(&cnt, 1u); if (pos < list_cap) list[pos] = (u32)k; }
    }
    __syncthreads();
    if (threadIdx.x == 0) *n_out = cnt < list_cap ? cnt : list_cap;
}

// ---- variant 1 only: compaction of the global edge table in ascending slot order -------------------
#define K2_TILE 2048   // table slots per workgroup (256 threads x 8)

__global__ __launch_bounds__(256) void k2_edge_count(Dev d) {
    const u32 tile = blockIdx.x;
    const u64* __restrict__ k = d.ekeys + (size_t)tile * K2_TILE + threadIdx.x * 8;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) c += k[j] != SG_EKEY_EMPTY;
    c = wave_sum_u32(c);
    __shared__ u32 s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) d.tile_cnt[tile] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(1024) void k2_scan_tiles(Dev d, u32 ntiles) {
    __shared__ u32 wsum[17];
    const u32 per = (ntiles + 1023) / 1024;
    const u32 beg = threadIdx.x * per < ntiles ? threadIdx.x * per : ntiles, end = beg + per < ntiles ? beg + per : ntiles;
    u32 c = 0;
    for (u32 i = beg; i < end; i++) c += d.tile_cnt[i];
    u32 total;
    u32 run = block_excl_scan<1024>(c, wsum, &total);
    for (u32 i = beg; i < end; i++) { const u32 v = d.tile_cnt[i]; d.tile_off[i] = run; run += v; }
    if (threadIdx.x == 0) {
        d.ctr[C_EDGES_FOUND] = total;
        if ((u64)total > d.max_edges) d.ctr[C_DROPPED_CAP] += (u64)total - d.max_edges;
    }
}

__global__ __launch_bounds__(256) void k2_edge_compact(Dev d) {
    const u32 tile = blockIdx.x;
    if (d.tile_cnt[tile] == 0) return;
    const u32 nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS], nob = (u32)d.ctr[C_N_OBIP];
    const u32 base_slot = tile * K2_TILE + threadIdx.x * 8;
    u64 k[8]; u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { k[j] = d.ekeys[(size_t)base_slot + j]; c += k[j] != SG_EKEY_EMPTY; }
    __shared__ u32 wsum[4];
    u32 incl = c;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const u32 o = __shfl_up(incl, s, 64); if ((int)(threadIdx.x & 63) >= s) incl += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < (threadIdx.x >> 6); w++) woff += wsum[w];
    u64 pos = (u64)d.tile_off[tile] + woff + incl - c;
#pragma unroll
    for (int j = 0; j < 8; j++) if (k[j] != SG_EKEY_EMPTY) {
        if (pos < d.max_edges) {
            const u32 f = dense_of(d, (u32)(k[j] >> 32), nk, nl, nob), t = dense_of(d, (u32)k[j], nk, nl, nob);
            d.e_slot[pos] = base_slot + j; d.e_from[pos] = f; d.e_to[pos] = t;
            if (f != SG_NONE && t != SG_NONE) atomicAdd(&d.deg[SG_DEG_IDX(f, tile & (SG_DEG_REP - 1))], 1u);
            else atomicAdd(&d.ctr[C_DROPPED_CAP], d.eacc[(size_t)(base_slot + j) * 4] & 0xFFFFFFFFull);
        }
        pos++;
    }
}

// rowptr = exclusive scan of the row degrees; rowptr[N] = E = edges of the window.  Multi-workgroup, single
// pass: workgroup b owns rows [b*1024, (b+1)*1024): eight lanes per row read its SG_DEG_REP replica counters
// (one sector each, consecutive lanes -> consecutive sectors), turn them into offsets inside the row and
// give the row's degree; a block scan makes local row offsets; the sum of the preceding workgroups' totals
// comes from rp_tot[] (each workgroup publishes (epoch, total) as soon as it knows it and the later ones
// wait for it — up to SG_LB_RESIDENT workgroups all are resident; larger grids order themselves by ticket, see below).
// Row degrees and the edges' positions inside their rows without device atomics (Dev::dh_g workgroups; see sg_device.h).  Workgroup g
// owns the output partitions [g * dh_ppw, (g + 1) * dh_ppw): it counts their sources in an LDS array indexed by node — the value a
// returning LDS add hands back is the edge's rank among the edges (g, source), stored to e_rank with a coalesced write — and publishes
// the array as row g of dh_hist.  K2_DH_FLIGHT source loads per thread are in flight together (a partition's ~1000 sources are one
// round trip for a 1024-thread workgroup: taken one partition at a time the kernel would be dh_ppw dependent round trips).
#define K2_DH_THREADS 1024
#define K2_DH_FLIGHT 16
#define K2_DH_GMAX 128           // k2_rowptr keeps a row's column of counts in registers: GMAX / 8 per lane
#define SG_WARM_WINDOW(d) ((d).warm && !(d).ctr[C_COLD])             /* (uniform) this window is closed on the warm path: the rebuild kernels return at once */
// The rebuild chain (k2_rowptr .. k2_rowsort_gather) on an engine that keeps state runs in one of three ways, decided on the device:
//   0  a full rebuild (cold window) on the Dev the host set up — its CSR pointers are the KEPT arrays (buffer 0);
//   1  a warm window that met NEW edges (C_DELTA_N != 0): the same kernels on those few edges only — the warm pass B left them in the
//      partition outputs, ranks from deg2 — and the result is the DELTA CSR (dc_*), which kw_compact merges in;
//  -1  a warm window without new edges: nothing to do, return at once.
// COMPACT node ids (round 6).  The kept CSR used to hold DENSE ids — known ids, then labels from N_KNOWN on — so every new pod moved the
// labels' ids and cost a full rebuild.  It holds compact indices now (known id | max_known + label: what the key mix works on): they never
// move, they order exactly as the dense ids do (dense = c below max_known, N_KNOWN + (c - max_known) above: monotone), and kw_compact maps
// them when it writes the window's CSR.  Raw outbound IPs have no such index (theirs is a slot of the window's own table; their dense ids
// are ranks among the window's): a window that has any is built in dense ids as before and leaves the kept state invalid.
__device__ __forceinline__ bool sg_kept_compact(const Dev& d) { return d.kept_compact && d.ctr[C_N_OBIP] == 0; }
__device__ __forceinline__ u32 sg_chain_rows(const Dev& d) { return sg_kept_compact(d) ? d.max_known + d.max_labels : (u32)d.ctr[C_N_NODES]; }   // rows of the CSR a chain launch builds
__device__ __forceinline__ int sg_chain_mode(const Dev& d) { if (!d.warm || d.ctr[C_COLD]) return 0; return d.ctr[C_DELTA_N] ? 1 : -1; }
__device__ __forceinline__ Dev sg_delta_view(const Dev& d) { Dev x = d; x.rowptr = d.dc_rowptr; x.col = d.dc_col; x.csr_from = d.dc_from; x.acc_csr = d.dc_acc; x.deg = d.deg2; return x; }
__global__ __launch_bounds__(K2_DH_THREADS) void k2_deg_hist(Dev d) {
    extern __shared__ u32 dh_cnt[];                                  // [N]
    if (SG_WARM_WINDOW(d)) return;
    const u32 N = (u32)d.ctr[C_N_NODES], g = blockIdx.x, t = threadIdx.x;
    const u32 CH = (d.pcap + K2_DH_THREADS - 1) / K2_DH_THREADS, items = d.dh_ppw * CH;   // work item = 1024 consecutive slots of one partition
    // (the first round's loads are issued before the counters are cleared: they fly while the LDS is zeroed)
    u32 fv[K2_DH_FLIGHT], sl[K2_DH_FLIGHT], okm = 0;
    auto issue = [&](const u32 it0) {
        okm = 0;
#pragma unroll
        for (int q = 0; q < K2_DH_FLIGHT; q++) {
            const u32 it = it0 + (u32)q < items ? it0 + (u32)q : items - 1;   // (uniform)
            const u32 k = it / CH, c = it - k * CH, oq = g * d.dh_ppw + k, i = c * K2_DH_THREADS + t;
            // (the source is loaded whether or not the slot holds an edge of this window — what lies beyond the partition's count is an
            // older window's node id, ignored below: the loads do not wait for the counts' round trip)
            sl[q] = oq * d.pcap + (i < d.pcap ? i : 0u);             // (slots: npb * pcap < 2^32 — the host sees to it)
            fv[q] = d.e_from[sl[q]];
            const bool ok = it0 + (u32)q < items && i < d.part_n[oq];
            okm |= ok ? (1u << q) : 0u;
        }
    };
    issue(0);
    for (u32 i = t; i < N; i += K2_DH_THREADS) dh_cnt[i] = 0;
    __syncthreads();
    for (u32 it0 = 0; it0 < items; it0 += K2_DH_FLIGHT) {
        if (it0) issue(it0);
#pragma unroll
        for (int q = 0; q < K2_DH_FLIGHT; q++) if (((okm >> q) & 1u) && fv[q] < N) d.e_rank[sl[q]] = atomicAdd(&dh_cnt[fv[q]], 1u);
    }
    __syncthreads();
    u32* out = d.dh_hist + (size_t)g * d.dh_ns;
    for (u32 i = t; i < N; i += K2_DH_THREADS) out[i] = dh_cnt[i];
}

// (RPR rows per workgroup.  DH — Dev::dh_g: the degrees come from k2_deg_hist's counts, 64 rows per workgroup so that the whole chip
// pulls the [dh_g][N] table; otherwise from the replica counters pass B's device atomics left, 256 rows per workgroup.)
#define K2_RP_ROWS_DH 64
template <u32 RPR, bool DH>
__global__ __launch_bounds__(1024) void k2_rowptr(Dev dd, u32 epoch) {
    __shared__ u32 wsum[17];
    __shared__ u32 rdeg[RPR];
    __shared__ u32 nlong, lbase, pre, bdyn;
    const int cm = sg_chain_mode(dd);
    if (cm < 0) return;
    const bool delta = cm == 1;
    const Dev d = delta ? sg_delta_view(dd) : dd;
    const u32 t = threadIdx.x;
    // Which rows this workgroup owns.  The look-back below waits for the workgroups of the rows before it.  Up to SG_LB_RESIDENT
    // workgroups (one per CU) every workgroup of the launch is resident and the block index serves.  Beyond that (C5: 150 k rows) a
    // workgroup takes its index from a ticket counter instead, so that it only ever waits for workgroups that have already STARTED —
    // HIP does not promise that blocks are dispatched in index order.  The counter resets itself: the workgroup that draws
    // the launch's last ticket is the last one to draw.
    u32 b = blockIdx.x;
    if (gridDim.x > SG_LB_RESIDENT) {                                // (uniform)
        if (t == 0) { const u32 tk = atomicAdd(&d.lb_ticket[0], 1u); if (tk == gridDim.x - 1) atomicExch(&d.lb_ticket[0], 0u); bdyn = tk; }
        __syncthreads();
        b = bdyn;
    }
    const u32 r0 = b * RPR;
    // (DH: the counts are loaded before N is known — rows beyond it read stale words inside the table (its rows are ncap + 1 rounded up
    // to 64 words) and are zeroed below: one dependent round trip less)
    constexpr u32 GLd = DH ? K2_DH_GMAX / 16 : 1;
    u32 v[GLd];
    if constexpr (DH) {
        const u32 GG = d.dh_g >> 4, rep = t >> 6, row = r0 + (t & 63u);
#pragma unroll
        for (u32 j = 0; j < GLd; j++) v[j] = j < GG ? d.dh_hist[(size_t)(rep * GG + j) * d.dh_ns + row] : 0u;
    }
    const u32 N = sg_chain_rows(d);
    if (r0 >= N && b != 0) return;                                   // beyond the last row (grid sized for ncap)
    if (t == 0) nlong = 0;
    // 1. replicas -> in-row offsets, row degrees
    if constexpr (DH) {
        // k2_deg_hist's counts: row r's column dh_hist[0 .. dh_g)[r] becomes its exclusive prefix (the offset of the edges (g, r) inside
        // row r), the total the row's degree.  Lane = (row, sixteenth of the column): a wave = 64 adjacent rows (coalesced reads and
        // writes), the sixteenths are the workgroup's sixteen waves — their totals meet in LDS.  The offsets are stored at once: the
        // stores fly under the scan and the wait for the preceding workgroups' totals.
        static_assert(RPR == 64, "one row per lane of a wave");
        constexpr u32 GL = K2_DH_GMAX / 16;
        __shared__ u32 dtot[16][RPR];
        const u32 GG = d.dh_g >> 4, rep = t >> 6, rl = t & 63u, row = r0 + rl;   // (dh_g: a multiple of 16, <= K2_DH_GMAX)
#pragma unroll
        for (u32 j = 0; j < GL; j++) v[j] = row < N ? v[j] : 0u;
        u32 run = 0;
#pragma unroll
        for (u32 j = 0; j < GL; j++) { const u32 x = v[j]; v[j] = run; run += x; }
        dtot[rep][rl] = run;
        __syncthreads();
        u32 before = 0, tot = 0;
#pragma unroll
        for (u32 r = 0; r < 16; r++) { const u32 x = dtot[r][rl]; before += r < rep ? x : 0u; tot += x; }
#pragma unroll
        for (u32 j = 0; j < GL; j++) if (j < GG && row < N) d.dh_hist[(size_t)(rep * GG + j) * d.dh_ns + row] = v[j] + before;
        if (rep == 0) rdeg[rl] = tot;
    } else {
    static_assert(DH || RPR % 128 == 0, "eight lanes per row, 128 rows per pass");
    for (u32 pass = 0; pass < RPR / 128; pass++) {
        const u32 rl = pass * 128 + (t >> 3), row = r0 + rl, rep = t & 7;
        const u32 dv = row < N ? d.deg[SG_DEG_IDX(row, rep)] : 0u;
        u32 incl = dv;                                               // inclusive prefix over the 8 lanes of the row
#pragma unroll
        for (int s2 = 1; s2 < 8; s2 <<= 1) { const u32 o = __shfl_up(incl, s2, 8); if ((int)rep >= s2) incl += o; }
        if (row < N) d.deg[SG_DEG_IDX(row, rep)] = incl - dv;
        if (rep == 7) rdeg[rl] = incl;
    }
    }
    __syncthreads();
    // 2. local scan
    const u32 dg = t < RPR ? rdeg[t] : 0u;
    u32 total;
    const u32 run = block_excl_scan<1024>(dg, wsum, &total);
    // 3. totals of the preceding workgroups
    if (t == 0) {
        // (relaxed, device scope: the word carries everything its readers want — (epoch, total) — so nothing has to be ordered before
        // it; a release store here waited for the workgroup's own stores and wrote the L2 back, and every acquire load of the poll loop
        // below invalidated it: per workgroup, 236 times)
        __hip_atomic_store(&d.rp_tot[b], ((u64)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pre = 0;
    }
    __syncthreads();
    {
        u32 mine = 0;
        for (u32 j = t; j < b; j += 1024) {
            u64 x;
            do { x = __hip_atomic_load(&d.rp_tot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((u32)(x >> 32) != epoch);
            mine += (u32)x;
        }
        if (b) { mine = wave_sum_u32(mine); if ((t & 63) == 0 && mine) atomicAdd(&pre, mine); }
    }
    __syncthreads();
    const u32 base = pre;
    // 4. publish.  Row starts are CLAMPED to the edge capacity: when a window holds more distinct edges than max_edges (counted:
    // C_DROPPED_CAP below), the rows behind the capacity are empty and the row across it is cut — every consumer of rowptr
    // (row sort, gather, in-statistics, alive marks) then stays inside the max_edges-sized arrays without clamping of its own.
    // (Unclamped, K4's gather walked d.col up to E_found: with 10 M events of ten different traces in one C2-sized window that
    // was a GPU memory fault.)
    const u64 s0u = (u64)base + run, s1u = s0u + (t < RPR ? rdeg[t] : 0u);
    const u32 s0 = (u32)(s0u < d.max_edges ? s0u : d.max_edges), s1 = (u32)(s1u < d.max_edges ? s1u : d.max_edges);
    const u32 dgc = s1 - s0;                                         // the row's edges inside the capacity
    if (t < RPR && r0 + t < N) {
        d.rowptr[r0 + t] = s0;
        if (dgc > 64) atomicAdd(&nlong, 1u);
        if (dgc > SG_MEAN_BLOCK) {                                   // a hub row: one work item per 512-neighbour block (k4_gather spreads them over the chip)
            const u32 nblk = (dgc + SG_MEAN_BLOCK - 1) / SG_MEAN_BLOCK;
            const u32 ib = (u32)atomicAdd(&d.ctr[C_HUB_ITEMS], (u64)nblk);   // C_HUB_ITEMS is zeroed by kc_prepare
            d.hub_base[r0 + t] = ib;
            for (u32 j = 0; j < nblk; j++) if (ib + j < d.hub_cap) d.hub_items[ib + j] = make_uint2(r0 + t, j);
        }
    }
    __syncthreads();
    if (t == 0) lbase = nlong ? (u32)atomicAdd(&d.ctr[C_N_LONG], (u64)nlong) : 0u;   // C_N_LONG is zeroed by kc_prepare
    __syncthreads();
    {   // positions inside this workgroup's slice of the long-row list
        __shared__ u32 lpos;
        if (t == 0) lpos = 0;
        __syncthreads();
        if (t < RPR && r0 + t < N && dgc > 64) d.longrows[lbase + atomicAdd(&lpos, 1u)] = r0 + t;
    }
    if (t == 0) {
        if (b == 0) {
            d.ctr[C_OVF_N] = 0;                                        // K1b has consumed the overflow list
            d.ctr[C_ACT_L] = SG_ACT_NONE; d.ctr[C_ACT_P] = 0;          // no active lists yet for this window (see k6_active_lists)
        }
        if (r0 + RPR >= N) {                                    // the workgroup of the last row knows E
            const u32 E = base + total;
            d.rowptr[N] = (u64)E < d.max_edges ? E : (u32)d.max_edges;
            if (!delta) {                                            // (the delta CSR's size is its last row pointer; the window's counts are kw_compact's)
            d.ctr[C_N_EDGES] = (u64)E < d.max_edges ? E : d.max_edges;
            if (d.variant == 0) { d.ctr[C_EDGES_FOUND] = E; if ((u64)E > d.max_edges) d.ctr[C_DROPPED_CAP] += (u64)E - d.max_edges; }
            if (d.warm) { d.ctr[C_KEPT_E] = (u64)E < d.max_edges ? E : d.max_edges; d.ctr[C_KEPT_BUF] = 0; }   // (this launch rebuilt the KEPT CSR, buffer 0: kw_compact, next, walks that many positions)
            }
        }
    }
}

// scatter into CSR rows (order inside a row is fixed afterwards by the row sort)
__global__ __launch_bounds__(256) void k2_scatter_table(Dev d) {
    const u64 found = d.ctr[C_EDGES_FOUND] < d.max_edges ? d.ctr[C_EDGES_FOUND] : d.max_edges;
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < found; i += gridDim.x * 256) {
        const u32 f = d.e_from[i];
        if (f == SG_NONE || d.e_to[i] == SG_NONE) {                    // endpoint beyond max_outbound_ips: dropped; clear its table slot
            const u32 sl = d.e_slot[i];
            d.ekeys[sl] = SG_EKEY_EMPTY;
            ulonglong2* a = reinterpret_cast<ulonglong2*>(d.eacc + (size_t)sl * 4); a[0] = make_ulonglong2(0, 0); a[1] = make_ulonglong2(0, 0);
            if (d.hist) { uint4* hs = reinterpret_cast<uint4*>(d.hist_src + (size_t)sl * SG_HIST_BINS); const uint4 z = make_uint4(0, 0, 0, 0); hs[0] = z; hs[1] = z; hs[2] = z; hs[3] = z; }
            continue;
        }
        const u32 pos = d.rowptr[f] + atomicAdd(&d.cursor[f], 1u);
        d.cs[pos] = make_uint2(d.e_to[i], d.e_slot[i]);
    }
}
__global__ __launch_bounds__(256) void k2_scatter_parts(Dev dd) {
    const int cm = sg_chain_mode(dd);
    if (cm < 0) return;
    const Dev d = cm == 1 ? sg_delta_view(dd) : dd;
    const u32 p = blockIdx.x, n = d.part_n[p];
    if (cm == 1 && n == 0) return;                                   // (the new edges of a warm window: most partitions have none)
    // four edges per thread and trip: their loads (source, then row start + replica offset) are in flight together — one edge per
    // trip was two dependent round trips for each of a partition's ~4 edges per thread.  The slots are read whether or not they hold an
    // edge of this window (beyond the count: an older window's node ids — valid indices, ignored at the store): the first trip's loads
    // do not wait for the count.
    const u32 pc = d.pcap;
    for (u32 i0 = threadIdx.x; i0 < pc && ((i0 < 1024u && cm != 1) || i0 < n); i0 += 1024) {
        u32 f[4], to[4], rk[4], slot[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 i = i0 + 256u * q;
            slot[q] = p * pc + (i < pc ? i : i0);
            f[q] = d.e_from[slot[q]]; to[q] = d.e_to[slot[q]]; rk[q] = d.e_rank[slot[q]];
        }
        u32 rp[4], dg[4];
        if (cm == 1) {                                               // the new edges of a warm window: a row's edges take their places from a cursor (any order: the row sort follows)
#pragma unroll
            for (int q = 0; q < 4; q++) if (i0 + 256u * q < n) { const u32 pos = d.rowptr[f[q]] + atomicAdd(&d.cursor[f[q]], 1u); if (pos < d.max_edges) d.cs[pos] = make_uint2(to[q], slot[q]); }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            rp[q] = d.rowptr[f[q]];
            dg[q] = d.dh_g ? d.dh_hist[(size_t)(p / d.dh_ppw) * d.dh_ns + f[q]] : d.deg[SG_DEG_IDX(f[q], p & (SG_DEG_REP - 1))];   // offset of (p's group | replica, source) inside the row
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 pos = (u64)rp[q] + dg[q] + rk[q];              // row + replica offset + arrival order
            if (i0 + 256u * q < n && pos < d.max_edges) d.cs[pos] = make_uint2(to[q], slot[q]);   // (one scattered 8-byte write: the cost is per write request, not per byte)
        }
    }
}

// ---- row sort + gather: per row, sort by destination, move the accumulators into CSR order,
// reduce the row's out-statistics (plain reduction: one owner per row) and compute e_uv, lat_z,
// err_ratio of its edges.  Rows up to 64 edges are handled by one wave, 4 rows per workgroup at a time.
__device__ __forceinline__ double mean_us(u64 sum_ns, u64 cnt) { return cnt ? ((double)sum_ns / 1000.0) / (double)cnt : 0.0; }
__device__ __forceinline__ double std_us(u64 sum_ns, u64 ssq_us, u64 cnt) {
    if (!cnt) return 0.0;
    const double m = mean_us(sum_ns, cnt);
    const double v = (double)ssq_us / (double)cnt - m * m;
    return v > 0.0 ? sqrt(v) : 0.0;
}

// What the row sort does per edge: move the accumulators into CSR order and (variant 1) free the table slot.
// The fp32 edge features, lat_z and err_ratio are computed afterwards, one thread per edge, by the edge
// workgroups of k3_node_features: inside the row sort they were ~1000 fp64-heavy instructions per edge run
// by the few threads that own a long row (a 3000-edge hub row kept one 256-thread workgroup busy for tens of us).
struct EdgeEmitArgs { u64* acc_csr; u32* csr_from; u64* eacc; u64* ekeys; u32* alive_csr; u32 variant; u32* hist_src; u32* hist_csr; u32 hist; u32* pos_of_slot; u32* slot_of_pos; };
__device__ __forceinline__ void edge_emit(const EdgeEmitArgs d, u32 pos, u32 row, u32 slot, u64, u64, u64, const ulonglong2 x, const ulonglong2 y) {
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(d.acc_csr + (size_t)pos * 4);
    dst[0] = x; dst[1] = y;
    d.csr_from[pos] = row;
    if (d.pos_of_slot) d.pos_of_slot[slot] = pos;                    // warm windows: where the edge of this partition-output slot sits in the CSR (kw_capture)
    if (d.slot_of_pos) d.slot_of_pos[pos] = slot;                    // delta windows: the partition-output slot of a delta position (kw_compact finds the edge's image index through it)
    d.alive_csr[pos] = 0;                                            // k3_in_stats adds the window's open connections
    if (d.hist) {                                                    // f-3: the edge's latency histogram follows it into row order
        uint4* hs = reinterpret_cast<uint4*>(d.hist_src + (size_t)slot * SG_HIST_BINS); uint4* hd = reinterpret_cast<uint4*>(d.hist_csr + (size_t)pos * SG_HIST_BINS);
        const uint4 h0 = hs[0], h1 = hs[1], h2 = hs[2], h3 = hs[3];
        hd[0] = h0; hd[1] = h1; hd[2] = h2; hd[3] = h3;
        if (d.variant == 1) { const uint4 z = make_uint4(0, 0, 0, 0); hs[0] = z; hs[1] = z; hs[2] = z; hs[3] = z; }   // ... and the table's bins are re-armed
    }
    if (d.variant == 1) {                                            // variant 1: this is also the window reset of the edge table
        ulonglong2* src = reinterpret_cast<ulonglong2*>(d.eacc + (size_t)slot * 4);
        src[0] = make_ulonglong2(0, 0); src[1] = make_ulonglong2(0, 0);
        d.ekeys[slot] = SG_EKEY_EMPTY;
    }
}
// (float)log1p((double)c) for an integer count: from the table the device itself filled with the same expression (bit-identical
// by construction), the fp64 log1p only beyond it
// sg_clock_probe: every CU spins on dependent integer VALU work for `iters` trips; workgroup 0 reports shader cycles and
// 100 MHz ticks of the same interval (their ratio x 100 = the shader clock in MHz the chip sustains under an all-CU load)
__global__ __launch_bounds__(256) void k_clock_spin(u64* clk, u32 iters) {
    u32 a = threadIdx.x, b = a * 3u + 1u, c = a * 5u + 7u;
    const u64 c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (u32 i = 0; i < iters; i++) { a = a * 1664525u + b; b = b * 22695477u + c; c ^= a >> 3; }
    const u64 c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[2] = c1 - c0; clk[3] = r1 - r0; }
    if ((a ^ b ^ c) == 0x12345678u) clk[3] = a;                      // (keeps the loop)
}
// latency probe (sg_latency_probe): word 0 of line x holds the next line, (A x + C) mod lines — a full-period walk for lines = 2^k
#define SG_CHASE_A 0x9E3779B5u
#define SG_CHASE_C 0x7F4A7C15u
__global__ __launch_bounds__(256) void k_chase_init(u32* buf, u32 mask) {
    for (u64 x = (u64)blockIdx.x * 256 + threadIdx.x; x <= mask; x += (u64)gridDim.x * 256) buf[x * 32] = ((u32)x * SG_CHASE_A + SG_CHASE_C) & mask;
    if (blockIdx.x == 0 && threadIdx.x == 0) buf[1] = mask;          // (word 1 of line 0: the walk's mask, for the many-chain launch's starting points)
}
__device__ __forceinline__ u32 out_mask(const u32* buf) { return buf[1]; }
__global__ void k_chase(const u32* buf, u32 steps, u64* out) {
    // one chain per LANE: a launch of 1 x 1 measures the unloaded latency; 1024 x 64 lanes keep 65 536 dependent chains in flight (every
    // lane starts somewhere else on the same full-period walk) — the latency of a random 128-byte line while the memory system is busy,
    // which is where the boxes of the pool differ
    const u32 mask_start = (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B1u;
    u32 x = (gridDim.x * blockDim.x) == 1 ? 0u : (mask_start & out_mask(buf));
    const u64 t0 = wall_clock64();
    for (u32 i = 0; i < steps; i++) x = __builtin_nontemporal_load(buf + (size_t)x * 32);   // (each address comes out of the load before it)
    const u64 t1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
    if (x == 0xFFFFFFFFu) out[1] = x;                                // (keeps the chain)
}
__global__ void k_l1p_table(float* tab) { const u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < SG_L1P_TAB) tab[i] = (float)log1p((double)i); }
__device__ __forceinline__ float log1p_count(const Dev& d, u64 c) { return c < SG_L1P_TAB ? d.l1p_tab[c] : (float)log1p((double)c); }
// fp64 arithmetic for the per-edge features, where one result per edge is wanted to fp32 accuracy and the chip's fp64 rate is the bound
// (k3_node_features' edge workgroups: seven IEEE divisions and three libm log1p per edge were ~700 fp64 instructions, 18 of the
// kernel's 25 us at C3).  sg_div: v_rcp_f64 + two Newton steps + one correction, <= 2 ulp (operands here are finite, positive and far from
// the exponent range's ends).  sg_log1p_pos (x >= 0, finite): log(1 + x) = e ln 2 + 2 atanh(s), s = (m - 1) / (m + 1) for 1 + x = m 2^e,
// m in [sqrt(1/2), sqrt(2)) — nine odd terms (|s| < 0.172: the tenth is below 3e-17) — plus the rounding of 1 + x put back, a short series
// below 1e-4; <= 2 ulp of the fp64 result against long-double log1p over 1e-12 .. 1e14 (tools/log1p_check.py: no fp32 result differs
// from (float)log1p(x) in six million samples).  The oracle's libm values are matched to the last fp32 bit except where the fp64
// value sits within ~1e-15 of a rounding boundary.
__device__ __forceinline__ double sg_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    e = fma(-d, r, 1.0); return fma(r, e, r);
}
__device__ __forceinline__ double sg_div(double n, double d) { const double r = sg_rcp(d), q = n * r; return fma(fma(-d, q, n), r, q); }
__device__ __forceinline__ double sg_log1p_pos(double x) {
    const double y = 1.0 + x;
    double m = __builtin_amdgcn_frexp_mant(y);                       // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(y);
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m; e = lo ? e - 1 : e;
    const double s = sg_div(m - 1.0, m + 1.0), s2 = s * s;
    double p = 1.0 / 19.0;
    p = fma(p, s2, 1.0 / 17.0); p = fma(p, s2, 1.0 / 15.0); p = fma(p, s2, 1.0 / 13.0); p = fma(p, s2, 1.0 / 11.0);
    p = fma(p, s2, 1.0 / 9.0); p = fma(p, s2, 1.0 / 7.0); p = fma(p, s2, 1.0 / 5.0); p = fma(p, s2, 1.0 / 3.0);
    const double logm = fma(2.0 * s * s2, p, 2.0 * s);
    const double c = (x - (y - 1.0)) * sg_rcp(y);                    // what 1 + x lost
    const double ed = (double)e;
    const double r = fma(ed, 6.93147180369123816490e-01, logm + fma(ed, 1.90821492927058770002e-10, c));
    const double sm = x * (1.0 - x * (0.5 - x * (1.0 / 3.0 - 0.25 * x)));
    return x < 1e-4 ? sm : r;
}
// e_uv, lat_z, err_ratio of edge `pos` from its accumulators and its row's out-statistics
__device__ __forceinline__ void edge_features(const Dev& d, u32 pos) {
    const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)pos * 4);
    const ulonglong2 x = a[0], y = a[1];
    const u32 from = d.csr_from[pos];
    const u64 cnt = x.x & 0xFFFFFFFFull, err = x.x >> 32, sum = x.y, mx = y.x, ssq = y.y;
    const double rc = cnt ? sg_rcp((double)cnt) : 0.0, dc = (double)cnt;
    // mean and standard deviation in us (mean_us / std_us with the division above: the features and lat_z take them to fp32)
    const double sum_us = (double)sum * 1e-3;
    double m_e = sum_us * rc; m_e = cnt ? fma(fma(-dc, m_e, sum_us), rc, m_e) : 0.0;
    double q_e = (double)ssq * rc; q_e = cnt ? fma(fma(-dc, q_e, (double)ssq), rc, q_e) : 0.0;
    const double var = q_e - m_e * m_e, s_e = var > 0.0 ? sqrt(var) : 0.0;
    const double mu = d.row_mu[from], sd = d.row_sd[from];           // mean_us / std_us of the row's out-statistics: computed once per row by the row sort
    const double z = sd > 1.0 ? sg_div(m_e - mu, sd) : m_e - mu;
    const float lat_z = (float)z;
    // err / cnt correctly rounded to fp32: for integers below 2^24 the fp32 division IS (float)((double) err / (double) cnt) (rounding
    // twice through a format of at least 2 x 24 + 2 bits is innocuous for a quotient); the fp64 division beyond
    const float err_ratio = !cnt ? 0.0f : ((cnt | err) < (1ull << 24) ? (float)(u32)err / (float)(u32)cnt : (float)((double)err / (double)cnt));
    const float zc = lat_z < -8.0f ? -8.0f : (lat_z > 8.0f ? 8.0f : lat_z);
    float4* e = reinterpret_cast<float4*>(d.efeat + (size_t)pos * SG_F_EDGE);
    e[0] = make_float4(log1p_count(d, cnt), (float)sg_log1p_pos(m_e * 1e-3), (float)sg_log1p_pos(s_e * 1e-3), (float)sg_log1p_pos((double)mx * 1e-6));
    e[1] = make_float4(err_ratio, log1p_count(d, err), zc * 0.125f, 1.0f);
    d.latz[pos] = lat_z; d.errr[pos] = err_ratio;
}

#define K2_SORT_LDS 4096         // words of each of the row sort's two LDS arrays — at least: the host sizes them (Dev::k2_sortw) so that a node bitmap fits, up to K2_SORT_LDS_MAX
#define K2_SORT_LDS_MAX 16384
#define K2_LONG_WGS 1024
#define K2_WAVE_ROW 512          // rows of up to this many edges are sorted by ONE wave (bitmap rank in a wave-private slice of the LDS arrays)
#define K2_WAVE_BW  1024         // ... when the node bitmap fits this many words (N <= 32768)
// One row sorted by the whole workgroup (rows of more than K2_WAVE_ROW edges, or node spaces beyond the wave-private bitmaps).
__device__ __forceinline__ void k2_row_wg(const Dev& d, const EdgeEmitArgs& ea, const u32 rr, u32* sk, u32* sv, u64 (*red)[4], u32* bsum, const u32 BW) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 b = d.rowptr[rr];
    u32 m = d.rowptr[rr + 1] - b;
    if ((u64)b + m > d.max_edges) m = b < d.max_edges ? (u32)(d.max_edges - b) : 0;
    if (m == 0) return;
    const uint2* in = d.cs + b; u32* key = d.col + b;              // in: {destination, slot} as scattered; key: the row's sorted destinations (output only)
    u64 cnt = 0, err = 0, sum = 0, ssq = 0, mx = 0;
    if (BW <= d.k2_sortw && m <= 1024) {
        // The common long row (65..1024 edges): bitmap rank as below, but every thread keeps its (<= 4) elements
        // and their accumulators in registers — one global round trip (the accumulator gather, issued before the
        // rank is known), no scratch arrays, three barriers.
        for (u32 w = threadIdx.x; w < BW; w += 256) sk[w] = 0;
        __syncthreads();
        u32 mk[4], mv[4]; ulonglong2 ax[4], ay[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 i = threadIdx.x + q * 256;
            const uint2 kv = in[i < m ? i : m - 1]; mk[q] = i < m ? kv.x : 0u; mv[q] = i < m ? kv.y : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 i = threadIdx.x + q * 256;
            if (i < m) { atomicOr(&sk[mk[q] >> 5], 1u << (mk[q] & 31)); const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)mv[q] * 4); ax[q] = a[0]; ay[q] = a[1]; }
            else { ax[q] = make_ulonglong2(0, 0); ay[q] = make_ulonglong2(0, 0); }
        }
        __syncthreads();
        {   // sv[w] = number of set bits in words [0, w)
            const u32 per = (BW + 255) / 256, w0 = threadIdx.x * per < BW ? threadIdx.x * per : BW, w1 = w0 + per < BW ? w0 + per : BW;
            u32 c = 0;
            for (u32 w = w0; w < w1; w++) c += __popc(sk[w]);
            u32 tot;
            u32 run = block_excl_scan<256>(c, bsum, &tot);
            for (u32 w = w0; w < w1; w++) { sv[w] = run; run += __popc(sk[w]); }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) { cnt += ax[q].x & 0xFFFFFFFFull; err += ax[q].x >> 32; sum += ax[q].y; ssq += ay[q].y; mx = ay[q].x > mx ? ay[q].x : mx; }
#pragma unroll
        for (int q = 0; q < 4; q++) if (threadIdx.x + q * 256 < m) {
            const u32 k = mk[q], r = sv[k >> 5] + __popc(sk[k >> 5] & ((1u << (k & 31)) - 1u));
            key[r] = k;
            edge_emit(ea, b + r, rr, mv[q], 0, 0, 0, ax[q], ay[q]);
        }
        cnt = wave_sum_u64(cnt); err = wave_sum_u64(err); sum = wave_sum_u64(sum); ssq = wave_sum_u64(ssq); mx = wave_max_u64(mx);
        if (lane == 0) { red[0][wave] = cnt; red[1][wave] = err; red[2][wave] = sum; red[3][wave] = ssq; red[4][wave] = mx; }
        __syncthreads();
        cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3]; err = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        sum = red[2][0] + red[2][1] + red[2][2] + red[2][3]; ssq = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        mx = red[4][0]; for (int w = 1; w < 4; w++) mx = red[4][w] > mx ? red[4][w] : mx;
    } else if (BW <= d.k2_sortw) {
        // Bitmap rank: the destinations of one row are distinct node ids < N, so setting bit `to` in an
        // N-bit LDS bitmap and counting the bits below it IS the sorted position — O(m + N/32) per row
        // instead of a comparison sort (a 3000-edge hub row cost ~100 us in the bitonic network).
        for (u32 w = threadIdx.x; w < BW; w += 256) sk[w] = 0;
        __syncthreads();
        // (every pass over the row takes four elements per thread and round: their loads are independent and in flight
        // together — one element per round made a 3 700-edge row cost five passes x 15 dependent round trips, 75-95 us)
        for (u32 i0 = 0; i0 < m; i0 += 1024) {
            u32 k4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const u32 i = i0 + threadIdx.x + q * 256; k4[q] = in[i < m ? i : m - 1].x; }
#pragma unroll
            for (int q = 0; q < 4; q++) if (i0 + threadIdx.x + q * 256 < m) atomicOr(&sk[k4[q] >> 5], 1u << (k4[q] & 31));
        }
        __syncthreads();
        {   // sv[w] = number of set bits in words [0, w)
            const u32 per = (BW + 255) / 256, w0 = threadIdx.x * per < BW ? threadIdx.x * per : BW, w1 = w0 + per < BW ? w0 + per : BW;
            u32 c = 0;
            for (u32 w = w0; w < w1; w++) c += __popc(sk[w]);
            u32 tot;
            u32 run = block_excl_scan<256>(c, bsum, &tot);
            for (u32 w = w0; w < w1; w++) { sv[w] = run; run += __popc(sk[w]); }
        }
        __syncthreads();
        for (u32 i0 = 0; i0 < m; i0 += 1024) {               // rank -> CSR position: destination, accumulators; the row totals on the way
            u32 k4[4], v4[4]; ulonglong2 x4[4], y4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const u32 i = i0 + threadIdx.x + q * 256; const uint2 kv = in[i < m ? i : m - 1]; k4[q] = kv.x; v4[q] = kv.y; }
#pragma unroll
            for (int q = 0; q < 4; q++) { const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)v4[q] * 4); x4[q] = a[0]; y4[q] = a[1]; }
#pragma unroll
            for (int q = 0; q < 4; q++) if (i0 + threadIdx.x + q * 256 < m) {
                const u32 k = k4[q], r = sv[k >> 5] + __popc(sk[k >> 5] & ((1u << (k & 31)) - 1u));
                key[r] = k;                                  // (input and output are different arrays: no scratch, no second pass)
                edge_emit(ea, b + r, rr, v4[q], 0, 0, 0, x4[q], y4[q]);
                cnt += x4[q].x & 0xFFFFFFFFull; err += x4[q].x >> 32; sum += x4[q].y; ssq += y4[q].y; mx = y4[q].x > mx ? y4[q].x : mx;
            }
        }
        cnt = wave_sum_u64(cnt); err = wave_sum_u64(err); sum = wave_sum_u64(sum); ssq = wave_sum_u64(ssq); mx = wave_max_u64(mx);
        if (lane == 0) { red[0][wave] = cnt; red[1][wave] = err; red[2][wave] = sum; red[3][wave] = ssq; red[4][wave] = mx; }
        __syncthreads();
        cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3]; err = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        sum = red[2][0] + red[2][1] + red[2][2] + red[2][3]; ssq = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        mx = red[4][0]; for (int w = 1; w < 4; w++) mx = red[4][w] > mx ? red[4][w] : mx;
    } else if (m <= 1024) {
        // rank sort: keys in LDS, every thread counts the smaller keys of its (<= 4) elements
        for (u32 i = threadIdx.x; i < m; i += 256) sk[i] = in[i].x;
        __syncthreads();
        u32 mk[4], mv[4], rk[4]; ulonglong2 ax[4], ay[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 i = threadIdx.x + q * 256;
            mk[q] = i < m ? sk[i] : 0xFFFFFFFFu; mv[q] = i < m ? in[i].y : 0u; rk[q] = 0;
            if (i < m) { const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)mv[q] * 4); ax[q] = a[0]; ay[q] = a[1]; }
            else { ax[q] = make_ulonglong2(0, 0); ay[q] = make_ulonglong2(0, 0); }
        }
        for (u32 j = 0; j < m; j++) {
            const u32 kj = sk[j];
#pragma unroll
            for (int q = 0; q < 4; q++) rk[q] += kj < mk[q];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { cnt += ax[q].x & 0xFFFFFFFFull; err += ax[q].x >> 32; sum += ax[q].y; ssq += ay[q].y; mx = ay[q].x > mx ? ay[q].x : mx; }
        cnt = wave_sum_u64(cnt); err = wave_sum_u64(err); sum = wave_sum_u64(sum); ssq = wave_sum_u64(ssq); mx = wave_max_u64(mx);
        if (lane == 0) { red[0][wave] = cnt; red[1][wave] = err; red[2][wave] = sum; red[3][wave] = ssq; red[4][wave] = mx; }
        __syncthreads();
        cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3]; err = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        sum = red[2][0] + red[2][1] + red[2][2] + red[2][3]; ssq = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        mx = red[4][0]; for (int w = 1; w < 4; w++) mx = red[4][w] > mx ? red[4][w] : mx;
#pragma unroll
        for (int q = 0; q < 4; q++) if (threadIdx.x + q * 256 < m) { key[rk[q]] = mk[q]; edge_emit(ea, b + rk[q], rr, mv[q], cnt, sum, ssq, ax[q], ay[q]); }
    } else {
        u32 np2 = 1; while (np2 < m) np2 <<= 1;
        u32* gk = sk; u32* gv = sv;
        if (m > d.k2_sortw) { gk = d.sort_k + 2 * (size_t)b; gv = d.sort_v + 2 * (size_t)b; }   // private padded slice of the global scratch
        for (u32 i = threadIdx.x; i < np2; i += 256) { const uint2 kv = in[i < m ? i : m - 1]; gk[i] = i < m ? kv.x : 0xFFFFFFFFu; gv[i] = i < m ? kv.y : 0; }
        __syncthreads();
        for (u32 k = 2; k <= np2; k <<= 1)
            for (u32 j = k >> 1; j > 0; j >>= 1) {
                for (u32 i = threadIdx.x; i < np2; i += 256) {
                    const u32 x = i ^ j;
                    if (x > i) {
                        const u32 a = gk[i], c = gk[x];
                        if ((a > c) == ((i & k) == 0)) { gk[i] = c; gk[x] = a; const u32 tt = gv[i]; gv[i] = gv[x]; gv[x] = tt; }
                    }
                }
                __syncthreads();
            }
        for (u32 i = threadIdx.x; i < m; i += 256) {
            const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)gv[i] * 4);
            const ulonglong2 x = a[0], y = a[1];
            cnt += x.x & 0xFFFFFFFFull; err += x.x >> 32; sum += x.y; ssq += y.y; mx = y.x > mx ? y.x : mx;
        }
        cnt = wave_sum_u64(cnt); err = wave_sum_u64(err); sum = wave_sum_u64(sum); ssq = wave_sum_u64(ssq); mx = wave_max_u64(mx);
        if (lane == 0) { red[0][wave] = cnt; red[1][wave] = err; red[2][wave] = sum; red[3][wave] = ssq; red[4][wave] = mx; }
        __syncthreads();
        cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3]; err = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        sum = red[2][0] + red[2][1] + red[2][2] + red[2][3]; ssq = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        mx = red[4][0]; for (int w = 1; w < 4; w++) mx = red[4][w] > mx ? red[4][w] : mx;
        for (u32 i = threadIdx.x; i < m; i += 256) {
            const u32 slot = gv[i];
            key[i] = gk[i];
            const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)slot * 4);
            const ulonglong2 x = a[0], y = a[1];
            edge_emit(ea, b + i, rr, slot, cnt, sum, ssq, x, y);
        }
    }
    if (threadIdx.x == 0) {
        u64* t = d.st_sum + (size_t)rr * SG_NODE_STAT_SUM_WORDS;
        t[ST_OUT_DEG] = m; t[ST_OUT_CNT] = cnt; t[ST_OUT_ERR] = err; t[ST_OUT_SUM] = sum; t[ST_OUT_SSQ] = ssq;
        d.st_max[(size_t)rr * 2] = mx;
        d.row_mu[rr] = mean_us(sum, cnt); d.row_sd[rr] = std_us(sum, ssq, cnt);
    }
    __syncthreads();
}
// One SG_MEAN_BLOCK-edge block of a row of more than K2_SPLIT_ROW edges (the hub work items k2_rowptr lists for k4_gather serve the
// row sort too): the workgroup builds the row's whole node bitmap (the keys are 8 bytes an edge, eight loads per thread in
// flight), but gathers, ranks and emits only its own block — a 3 750-edge row is eight workgroups x ~3 round trips instead of
// one workgroup x ~12 (such rows were the tail of the launch: ~20 us each, two or three in a row for an unlucky workgroup).
// The row's out-statistics are integer sums: every block adds its share with device atomics (st_sum / st_max are zero since
// the window reset); k3_in_reduce, two launches later, turns the totals into ST_OUT_DEG / row_mu / row_sd (k2_split_finish).
// (A "last block finishes the row" ticket needs a release / acquire fence per block: on this chip that is an L2 write-back —
// buffer_wbl2 — and ~600 of them made the launch 40 us SLOWER than the unsplit row sort.)
#define K2_SPLIT_ROW 1024
// how many hub work items the row sort may use: all of them, when the whole list was recorded and a node bitmap fits the LDS arrays
__device__ __forceinline__ u32 k2_split_items(const Dev& d) {
    const u32 BW = (sg_chain_rows(d) + 31) >> 5;
    return (d.ctr[C_HUB_ITEMS] <= d.hub_cap && BW <= d.k2_sortw && !SG_ABL(d, 0x800u)) ? (u32)d.ctr[C_HUB_ITEMS] : 0u;
}
__device__ __forceinline__ void k2_split_finish(const Dev& d, u32 tid, u32 nt) {
    const u32 H = k2_split_items(d);
    for (u32 it = tid; it < H; it += nt) {
        const uint2 x = d.hub_items[it];
        if (x.y != 0) continue;
        const u32 b = d.rowptr[x.x]; u32 m = d.rowptr[x.x + 1] - b;
        if (m <= K2_SPLIT_ROW) continue;
        if ((u64)b + m > d.max_edges) m = b < d.max_edges ? (u32)(d.max_edges - b) : 0;
        u64* t = d.st_sum + (size_t)x.x * SG_NODE_STAT_SUM_WORDS;
        const u64 tc = t[ST_OUT_CNT], ts = t[ST_OUT_SUM], tq = t[ST_OUT_SSQ];
        t[ST_OUT_DEG] = m;
        d.row_mu[x.x] = mean_us(ts, tc); d.row_sd[x.x] = std_us(ts, tq, tc);
    }
}
__device__ __forceinline__ void k2_row_block(const Dev& d, const EdgeEmitArgs& ea, const u32 rr, const u32 blk, u32* sk, u32* sv, u64 (*red)[4], u32* bsum, const u32 BW) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 b = d.rowptr[rr];
    u32 m = d.rowptr[rr + 1] - b;
    if ((u64)b + m > d.max_edges) m = b < d.max_edges ? (u32)(d.max_edges - b) : 0;
    const uint2* in = d.cs + b; u32* key = d.col + b;
    u64 cnt = 0, err = 0, sum = 0, ssq = 0, mx = 0;
    const u32 e0 = blk * SG_MEAN_BLOCK;
    if (e0 < m) {                                                    // (uniform)
        for (u32 w = threadIdx.x; w < BW; w += 256) sk[w] = 0;
        __syncthreads();
        for (u32 i0 = 0; i0 < m; i0 += 2048) {
            u32 k8[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const u32 i = i0 + threadIdx.x + q * 256; k8[q] = in[i < m ? i : m - 1].x; }
#pragma unroll
            for (int q = 0; q < 8; q++) if (i0 + threadIdx.x + q * 256 < m) atomicOr(&sk[k8[q] >> 5], 1u << (k8[q] & 31));
        }
        constexpr int QB = SG_MEAN_BLOCK / 256;
        u32 mk[QB], mv[QB]; ulonglong2 ax[QB], ay[QB];
#pragma unroll
        for (int q = 0; q < QB; q++) {                               // this block's elements and their accumulators: in flight across the scan
            const u32 i = e0 + threadIdx.x + q * 256;
            const uint2 kv = in[i < m ? i : m - 1]; mk[q] = kv.x; mv[q] = kv.y;
        }
#pragma unroll
        for (int q = 0; q < QB; q++) {
            if (e0 + threadIdx.x + q * 256 < m) { const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)mv[q] * 4); ax[q] = a[0]; ay[q] = a[1]; }
            else { ax[q] = make_ulonglong2(0, 0); ay[q] = make_ulonglong2(0, 0); }
        }
        __syncthreads();
        {   // sv[w] = number of set bits in words [0, w)
            const u32 per = (BW + 255) / 256, w0 = threadIdx.x * per < BW ? threadIdx.x * per : BW, w1 = w0 + per < BW ? w0 + per : BW;
            u32 c = 0;
            for (u32 w = w0; w < w1; w++) c += __popc(sk[w]);
            u32 tot;
            u32 run = block_excl_scan<256>(c, bsum, &tot);
            for (u32 w = w0; w < w1; w++) { sv[w] = run; run += __popc(sk[w]); }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QB; q++) if (e0 + threadIdx.x + q * 256 < m) {
            const u32 k = mk[q], r = sv[k >> 5] + __popc(sk[k >> 5] & ((1u << (k & 31)) - 1u));
            key[r] = k;
            edge_emit(ea, b + r, rr, mv[q], 0, 0, 0, ax[q], ay[q]);
            cnt += ax[q].x & 0xFFFFFFFFull; err += ax[q].x >> 32; sum += ax[q].y; ssq += ay[q].y; mx = ay[q].x > mx ? ay[q].x : mx;
        }
        cnt = wave_sum_u64(cnt); err = wave_sum_u64(err); sum = wave_sum_u64(sum); ssq = wave_sum_u64(ssq); mx = wave_max_u64(mx);
        if (lane == 0) { red[0][wave] = cnt; red[1][wave] = err; red[2][wave] = sum; red[3][wave] = ssq; red[4][wave] = mx; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u64* t = d.st_sum + (size_t)rr * SG_NODE_STAT_SUM_WORDS;
        if (e0 < m) {
            cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3]; err = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            sum = red[2][0] + red[2][1] + red[2][2] + red[2][3]; ssq = red[3][0] + red[3][1] + red[3][2] + red[3][3];
            mx = red[4][0]; for (int w = 1; w < 4; w++) mx = red[4][w] > mx ? red[4][w] : mx;
            if (cnt) atomicAdd(&t[ST_OUT_CNT], cnt);
            if (err) atomicAdd(&t[ST_OUT_ERR], err);
            if (sum) atomicAdd(&t[ST_OUT_SUM], sum);
            if (ssq) atomicAdd(&t[ST_OUT_SSQ], ssq);
            if (mx) atomicMax(&d.st_max[(size_t)rr * 2], mx);
        }
    }
    __syncthreads();
}
// One row of 65 .. K2_WAVE_ROW edges sorted by one wave, no barrier: the destinations of a row are distinct node ids < N, so
// setting bit `to` in an N-bit bitmap and counting the bits below it IS the sorted position.  bm / pf: the wave's private
// BW-word bitmap and word-prefix arrays (LDS operations of one wave execute in order).
__device__ __forceinline__ void k2_row_wave(const Dev& d, const EdgeEmitArgs& ea, const u32 rr, u32* bm, u32* pf, const u32 BW) {
    const u32 lane = threadIdx.x & 63;
    const u32 b = d.rowptr[rr];
    u32 m = d.rowptr[rr + 1] - b;
    if ((u64)b + m > d.max_edges) m = b < d.max_edges ? (u32)(d.max_edges - b) : 0;
    if (m == 0) return;
    const uint2* in = d.cs + b; u32* key = d.col + b;
    constexpr int Q = K2_WAVE_ROW / 64;
    u32 mk[Q], mv[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) { const u32 i = lane + 64u * q; const uint2 kv = in[i < m ? i : m - 1]; mk[q] = kv.x; mv[q] = kv.y; }
    for (u32 w = lane; w < BW; w += 64) bm[w] = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) if (lane + 64u * q < m) atomicOr(&bm[mk[q] >> 5], 1u << (mk[q] & 31));
    {   // pf[w] = set bits in words [0, w): lane l owns the words [l * per, (l + 1) * per)
        const u32 per = (BW + 63) >> 6, w0 = lane * per < BW ? lane * per : BW, w1 = w0 + per < BW ? w0 + per : BW;
        u32 c = 0;
        for (u32 w = w0; w < w1; w++) c += __popc(bm[w]);
        u32 incl = c;
        incl += dpp32<0x111>(incl); incl += dpp32<0x112>(incl); incl += dpp32<0x114>(incl); incl += dpp32<0x118>(incl);   // row_shr 1, 2, 4, 8
        const u32 r0 = rdlane32(incl, 15), r1 = rdlane32(incl, 31), r2 = rdlane32(incl, 47);
        incl += (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
        u32 run = incl - c;
        for (u32 w = w0; w < w1; w++) { pf[w] = run; run += __popc(bm[w]); }
    }
    u64 cnt = 0, err = 0, sum = 0, ssq = 0, mx = 0;
#pragma unroll
    for (int q0 = 0; q0 < Q; q0 += 4) {                              // four accumulator gathers in flight
        ulonglong2 x4[4], y4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)mv[q0 + q] * 4); x4[q] = a[0]; y4[q] = a[1]; }
#pragma unroll
        for (int q = 0; q < 4; q++) if (lane + 64u * (q0 + q) < m) {
            const u32 k = mk[q0 + q], r = pf[k >> 5] + __popc(bm[k >> 5] & ((1u << (k & 31)) - 1u));
            key[r] = k;
            edge_emit(ea, b + r, rr, mv[q0 + q], 0, 0, 0, x4[q], y4[q]);
            cnt += x4[q].x & 0xFFFFFFFFull; err += x4[q].x >> 32; sum += x4[q].y; ssq += y4[q].y; mx = y4[q].x > mx ? y4[q].x : mx;
        }
    }
    cnt = wave_sum_u64(cnt); err = wave_sum_u64(err); sum = wave_sum_u64(sum); ssq = wave_sum_u64(ssq); mx = wave_max_u64(mx);
    if (lane == 0) {
        u64* t = d.st_sum + (size_t)rr * SG_NODE_STAT_SUM_WORDS;
        t[ST_OUT_DEG] = m; t[ST_OUT_CNT] = cnt; t[ST_OUT_ERR] = err; t[ST_OUT_SUM] = sum; t[ST_OUT_SSQ] = ssq;
        d.st_max[(size_t)rr * 2] = mx;
        d.row_mu[rr] = mean_us(sum, cnt); d.row_sd[rr] = std_us(sum, ssq, cnt);
    }
}
// The same rows when the node space is too large for wave-private bitmaps (N > 32 768: a shard of config 5 has 150 k nodes, and
// every one of its ~200-edge rows took the whole workgroup, three barriers and a 256-thread rank loop — 0.85 ms of a 1.6 ms close):
// a rank sort inside the wave, independent of N.  Lane l keeps elements l, l + 64, ...; element j is broadcast with v_readlane (j is
// uniform) and every lane counts the keys below its own: m (1 + ceil(m / 64)) instructions per row, ~1 000 for a 200-edge row.
__device__ __forceinline__ void k2_row_wave_rank(const Dev& d, const EdgeEmitArgs& ea, const u32 rr) {
    const u32 lane = threadIdx.x & 63;
    const u32 b = d.rowptr[rr];
    u32 m = d.rowptr[rr + 1] - b;
    if ((u64)b + m > d.max_edges) m = b < d.max_edges ? (u32)(d.max_edges - b) : 0;
    if (m == 0) return;
    const uint2* in = d.cs + b; u32* key = d.col + b;
    constexpr int Q = K2_WAVE_ROW / 64;
    u32 mk[Q], mv[Q], rk[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) { const u32 i = lane + 64u * q; const uint2 kv = in[i < m ? i : m - 1]; mk[q] = i < m ? kv.x : 0xFFFFFFFFu; mv[q] = kv.y; rk[q] = 0; }
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const u32 nq = m > 64u * q ? (m - 64u * q < 64u ? m - 64u * q : 64u) : 0u;   // uniform
        for (u32 j = 0; j < nq; j++) {
            const u32 kj = rdlane32(mk[q], (int)j);
#pragma unroll
            for (int q2 = 0; q2 < Q; q2++) rk[q2] += kj < mk[q2];
        }
    }
    u64 cnt = 0, err = 0, sum = 0, ssq = 0, mx = 0;
#pragma unroll
    for (int q0 = 0; q0 < Q; q0 += 4) {                              // four accumulator gathers in flight
        if (64u * q0 >= m) break;                                    // uniform
        ulonglong2 x4[4], y4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)mv[q0 + q] * 4); x4[q] = a[0]; y4[q] = a[1]; }
#pragma unroll
        for (int q = 0; q < 4; q++) if (lane + 64u * (q0 + q) < m) {
            key[rk[q0 + q]] = mk[q0 + q];
            edge_emit(ea, b + rk[q0 + q], rr, mv[q0 + q], 0, 0, 0, x4[q], y4[q]);
            cnt += x4[q].x & 0xFFFFFFFFull; err += x4[q].x >> 32; sum += x4[q].y; ssq += y4[q].y; mx = y4[q].x > mx ? y4[q].x : mx;
        }
    }
    cnt = wave_sum_u64(cnt); err = wave_sum_u64(err); sum = wave_sum_u64(sum); ssq = wave_sum_u64(ssq); mx = wave_max_u64(mx);
    if (lane == 0) {
        u64* t = d.st_sum + (size_t)rr * SG_NODE_STAT_SUM_WORDS;
        t[ST_OUT_DEG] = m; t[ST_OUT_CNT] = cnt; t[ST_OUT_ERR] = err; t[ST_OUT_SUM] = sum; t[ST_OUT_SSQ] = ssq;
        d.st_max[(size_t)rr * 2] = mx;
        d.row_mu[rr] = mean_us(sum, cnt); d.row_sd[rr] = std_us(sum, ssq, cnt);
    }
}
__global__ __launch_bounds__(256) void k2_rowsort_gather(Dev dd) {
    const int cm = sg_chain_mode(dd);
    if (cm < 0) return;
    const Dev d = cm == 1 ? sg_delta_view(dd) : dd;
    const u32 N = sg_chain_rows(d), nlong = (u32)d.ctr[C_N_LONG];
    const EdgeEmitArgs ea = {d.acc_csr, d.csr_from, d.eacc, d.ekeys, d.alive_csr, d.variant, d.hist_src, d.hist_csr, d.hist, d.warm ? d.pos_of_slot : nullptr, cm == 1 ? d.dc_slot : nullptr};
    extern __shared__ u32 k2_lds[];                                  // 2 x k2_sortw words (dynamic: a node bitmap of the engine's node capacity fits when it can)
    u32* sk = k2_lds; u32* sv = k2_lds + d.k2_sortw;
    __shared__ u64 red[5][4];
    __shared__ u32 bsum[5];
    __shared__ u32 bigrow[4];
    const u32 BW = (N + 31) >> 5;                                    // words of a node bitmap
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- long rows (more than 64 edges): the first K2_LONG_WGS workgroups take four list entries at a time, a wave each; a row of
    // up to K2_WAVE_ROW edges is sorted by its wave alone (wave-private quarter of sk / sv), a longer one by the whole workgroup
    // afterwards (rows of 65..1024 edges used to cost a workgroup three barriers and a 256-thread scan of the bitmap EACH) ----
    const u32 nlw = gridDim.x > 2 * K2_LONG_WGS ? K2_LONG_WGS : (gridDim.x / 2 ? gridDim.x / 2 : 1);   // host launches >= 2 workgroups
    if (blockIdx.x < nlw) {
        const bool wave_ok = BW <= K2_WAVE_BW && !SG_ABL(d, 0x400u);
        // rows of more than K2_SPLIT_ROW edges first, a workgroup per 512-edge block (when the whole list was recorded and the bitmap fits)
        const u32 H = k2_split_items(d);
        for (u32 it = blockIdx.x; it < H; it += nlw) {
            const uint2 x = d.hub_items[it];
            if (d.rowptr[x.x + 1] - d.rowptr[x.x] > K2_SPLIT_ROW) k2_row_block(d, ea, x.x, x.y, sk, sv, red, bsum, BW);   // (uniform)
        }
        for (u32 l0 = blockIdx.x * 4; l0 < nlong; l0 += nlw * 4) {
            const u32 li = l0 + wave;
            u32 big = SG_NONE;
            if (li < nlong) {
                const u32 rr = d.longrows[li];
                const u32 m = d.rowptr[rr + 1] - d.rowptr[rr];
                if (wave_ok && m <= K2_WAVE_ROW) k2_row_wave(d, ea, rr, sk + wave * K2_WAVE_BW, sv + wave * K2_WAVE_BW, BW);
                else if (BW > K2_WAVE_BW && m <= K2_WAVE_ROW && !SG_ABL(d, 0x400u)) k2_row_wave_rank(d, ea, rr);
                else if (!(H && m > K2_SPLIT_ROW)) big = rr;
            }
            if (lane == 0) bigrow[wave] = big;
            __syncthreads();
            for (u32 w2 = 0; w2 < 4; w2++) {
                const u32 rr = bigrow[w2];
                if (rr == SG_NONE) continue;                         // uniform
                k2_row_wg(d, ea, rr, sk, sv, red, bsum, BW);
            }
            __syncthreads();
        }
        return;
    }
    // ---- rows of up to 64 edges: one wave per row ----
    for (u32 r = (blockIdx.x - nlw) * 4 + wave; r < N; r += (gridDim.x - nlw) * 4) {
        const u32 beg = d.rowptr[r];
        u32 n = d.rowptr[r + 1] - beg;
        if (n == 0 || n > 64) continue;
        if ((u64)beg + n > d.max_edges) n = beg < d.max_edges ? (u32)(d.max_edges - beg) : 0;
        if (n == 0) continue;
        const uint2 kv = d.cs[beg + (lane < n ? lane : 0u)];
        const u32 k = lane < n ? kv.x : 0xFFFFFFFFu, v = lane < n ? kv.y : 0;
        u32 rank = 0;
        for (u32 j = 0; j < n; j++) rank += rdlane32(k, (int)j) < k;   // j uniform: v_readlane
        ulonglong2 x = make_ulonglong2(0, 0), y = make_ulonglong2(0, 0);
        if (lane < n) { const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_src + (size_t)v * 4); x = a[0]; y = a[1]; }
        const u64 cnt = wave_sum_u64(x.x & 0xFFFFFFFFull), err = wave_sum_u64(x.x >> 32), sum = wave_sum_u64(x.y), ssq = wave_sum_u64(y.y), mx = wave_max_u64(y.x);
        if (lane < n) d.col[beg + rank] = k;
        if (lane == 0) {
            u64* t = d.st_sum + (size_t)r * SG_NODE_STAT_SUM_WORDS;
            t[ST_OUT_DEG] = n; t[ST_OUT_CNT] = cnt; t[ST_OUT_ERR] = err; t[ST_OUT_SUM] = sum; t[ST_OUT_SSQ] = ssq;
            d.st_max[(size_t)r * 2] = mx;
            d.row_mu[r] = mean_us(sum, cnt); d.row_sd[r] = std_us(sum, ssq, cnt);
        }
        if (lane < n) edge_emit(ea, beg + rank, r, v, cnt, sum, ssq, x, y);
    }
}

// ---- warm windows (sg_device.h): capture behind a full rebuild, one-pass window CSR on a warm window --------------------------------
// kw_capture, behind the rebuild of a COLD close (the first KW_CAPW workgroups of the kw_compact launch).  On an engine that keeps state the rebuild (k2_deg_hist .. k2_rowsort_gather) runs on a
// Dev whose CSR pointers are the KEPT arrays: it has just built the kept CSR — every key pass B's cold merge left in its tables, the
// window's own and the ones carried over from the old image — with each edge's accumulators (bit 63 of the max word = touched in this
// window).  Here every slot of the table image gets the kept position of its edge (image index -> partition-output index -> the
// position the row sort reported) and the state is declared whole, unless the window holds raw outbound IPs (their compact indices
// are slots of this window's own outbound-IP table, their node ids ranks among this window's own).  The scratch node statistics the
// rebuild wrote (it reduces every row it sorts; the window's real ones come from kw_compact) are zeroed for the next rebuild.
#define KW_CAPW 48                                                   // workgroups of the kw_compact launch that do this instead of a chunk (nothing in a chunk's work depends on it)
__device__ __forceinline__ void kw_capture(const Dev& d, u64* scratch_sum, u64* scratch_max, u32 wg, u32 nwg, u32 nthreads) {
    const u64 tid = (u64)wg * nthreads + threadIdx.x, nt = (u64)nwg * nthreads;
    if (!d.ctr[C_COLD]) {                                            // (uniform) a warm window changes nothing — unless its new edges went through the
        if (d.ctr[C_DELTA_N]) {                                      // row sort, which reduces every row it sorts into the scratch statistics: re-arm them
            for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_SUM_WORDS; i += nt) scratch_sum[i] = 0;
            for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_MAX_WORDS; i += nt) scratch_max[i] = 0;
        }
        return;
    }
    const bool whole = d.ctr[C_N_OBIP] == 0;
    if (tid == 0) {
        d.ctr[C_KEPT_VALID] = whole ? 1ull : 0ull;                   // (C_KEPT_E: k2_rowptr's count on the kept arrays)
        d.ctr[C_KEPT_NK] = d.ctr[C_N_KNOWN]; d.ctr[C_KEPT_NL] = d.ctr[C_N_LABELS];
        d.ctr[C_COLD_WINDOWS] += 1;
    }
    for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_SUM_WORDS; i += nt) scratch_sum[i] = 0;
    for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_MAX_WORDS; i += nt) scratch_max[i] = 0;
    const u64 slots = (u64)d.npb * d.k1b_ht;
    for (u64 i = tid; i < slots; i += nt) {
        const u32 oi = d.wk_pos[i];
        if (oi != SG_NONE) {
            const u32 pos = oi < d.pcap ? d.pos_of_slot[(size_t)(i / d.k1b_ht) * d.pcap + oi] : SG_NONE;
            d.wk_pos[i] = pos;
            if (pos != SG_NONE) d.k_slot[pos] = (u32)i;              // kept position -> image index (buffer 0: a full rebuild writes buffer 0), for the delta windows' renumbering
        }
    }
}

// kw_compact — every window of an engine that keeps state; the whole of K2 on a WARM one: the kept CSR minus the edges no record
// touched, in ONE stable pass (on a cold window the rebuild has just refreshed the kept CSR and kw_capture its positions).  Workgroup b owns
// the kept positions [b KW_CH, (b + 1) KW_CH): it loads their accumulators (written by the warm pass B, bit 63 of the max word =
// touched), columns and sources, counts the touched ones (wave ballots), publishes the count and sums the counts of the chunks before
// it (the look-back of k2_rowptr: relaxed (epoch, total) words; beyond SG_LB_RESIDENT chunks the workgroups order themselves by
// ticket), and writes the survivors at base + rank — adjacent lanes, adjacent addresses, the order inside every row unchanged, so
// rows stay sorted by destination.  The rows that START in the chunk get their new row pointer from the same ranks.  The out-
// statistics of a row (integer sums, order-free) are folded in LDS arrays indexed by row − first row of the chunk and leave with plain
// stores for the rows that lie wholly inside the chunk, with device atomics for the at most two that cross its ends (and for rows
// beyond the LDS arrays' reach in graphs of very short rows); k3_in_reduce turns the sums into degree, mean and deviation and lists
// the hub rows' blocks.
// Geometry (measured, C3, phase stamps: a chunk's workgroup lives ~12 us whatever its size — loads 3.7, scan + look-back 2.1, stores +
// folds 3.3, row pointers 2.1 — so the launch costs one such life per ROUND of workgroups): 512 threads x 4 positions = 2048 per
// chunk, 20 KiB of LDS, three workgroups per CU — C3's 565 working chunks are resident at once (1024 threads x 2048 positions:
// two per CU, 53 chunks in a second round, 40 us; 1024 x 4096 with 60 KiB: one per CU, 44 us).
#define KW_THREADS 512
#define KW_NW (KW_THREADS / 64)
#define KW_Q 4
#define KW_CH (KW_THREADS * KW_Q)
#define KW_ROWS 512                                                  // rows per chunk with LDS accumulators (5 x u64 each: 20 KiB)
#define KW_RESIDENT (3 * SG_LB_RESIDENT)                             // working chunks that are certainly resident together (<= 80 VGPRs, 21 KiB of LDS)

// kw_compact on a DELTA window (round 6): the warm pass B met keys the kept set lacks and emitted them as new edges; the rebuild chain has
// sorted them into the delta CSR (dc_rowptr / dc_col / dc_from / dc_acc, D edges).  The kept CSR (KE edges) and the delta CSR are two
// sorted lists of (source, destination) keys without a common key; their MERGE is the new kept CSR (KE + D positions, written to the
// other kept buffer: wk_pos follows through k_slot / dl_img), and the window's CSR is the merge minus the untouched kept edges.
// Workgroup b owns the MERGED positions [b KW_CH, (b + 1) KW_CH) — not kept positions: a window whose new edges all sort into one
// stretch of the kept order (a new pod's rows) had one chunk place fifty thousand of them, 246 us against 25 for its neighbours.  Two
// merge-path searches (how many kept keys are among the first m merged ones) give the chunk its kept range [i0, i1) and its delta range
// [j0, j1); the two short lists meet in LDS, every element finds its merged index by a binary search in the other list, a bit map of
// the touched elements in merged order gives the ranks, and the look-back over the chunks' touched counts the base — as on any window.
// Row statistics leave with atomics throughout (a row's elements may lie in two chunks whichever list they come from).
__device__ __forceinline__ bool kw_key_less(u32 af, u32 ac, u32 bf, u32 bc) { return af < bf || (af == bf && ac < bc); }
// kept keys among the first m of the merge (0 <= m <= KE + D): the smallest i in [max(0, m - D), min(m, KE)] whose kept key i is NOT below
// the new key m - i - 1.  Called by a whole WAVE: 64 candidates per step (a one-lane binary search is ~20 dependent trips to memory of
// four loads each — 15 us in front of every chunk of a delta window; 64-ary it is four).
__device__ __forceinline__ u32 kw_merge_path(const u32* kfrom, const u32* kcol, const u32* dfrom, const u32* dcol, u32 KE, u32 D, u32 m) {
    const u32 lane = threadIdx.x & 63u;
    u32 lo = m > D ? m - D : 0u, hi = m < KE ? m : KE;
    while (lo < hi) {                                                // (uniform)
        const u32 span = hi - lo;
        const u32 c = lo + (u32)(((u64)span * lane) >> 6);           // lo <= c < hi, ascending with the lane (repeats when span < 64)
        const u32 j = m - c;                                         // >= 1 (c < hi <= m), <= D (c >= lo >= m - D)
        const bool below = kw_key_less(kfrom[c], kcol[c], dfrom[j - 1], dcol[j - 1]);   // monotone: true up to some candidate, false from there on
        const u32 nt = (u32)__popcll(__ballot(below ? 1 : 0));
        const u32 nlo = nt ? (u32)__shfl((int)c, (int)nt - 1, 64) + 1u : lo;
        const u32 nhi = nt < 64u ? (u32)__shfl((int)c, (int)nt, 64) : hi;
        lo = nlo; hi = nhi;
    }
    return lo;
}
__device__ __forceinline__ void kw_compact_delta(const Dev& d, const u32 b, const u32 epoch, const u32 KE, const u32 N, const u32 D, const u32 buf,
                                                 u64* kw_racc, u32* pre) {
    __shared__ u32 kF[KW_CH], kC[KW_CH];                             // the chunk's keys: its kept ones [0, na), then its new ones [na, na + nd)
    __shared__ u32 tbits[KW_CH / 32];                                // touched, by merged index
    __shared__ u32 tpre[KW_CH / 32 + 1];                             // touched elements below word w
    __shared__ u32 dg[4];                                            // i0, i1 (merge path), then rows
    const u32 t = threadIdx.x, lane = t & 63u;
    const size_t KC = (size_t)d.npb * d.pcap;
    const u32* kcol = buf ? d.k_col2 : d.k_col; const u32* kfrom = buf ? d.k_from2 : d.k_from; const u32* krp = buf ? d.k_rowptr2 : d.k_rowptr;
    const u32* kslot = d.k_slot + (size_t)buf * KC;
    u32* ncol = buf ? d.k_col : d.k_col2; u32* nfrom = buf ? d.k_from : d.k_from2; u32* nrp = buf ? d.k_rowptr : d.k_rowptr2;
    u32* nslot = d.k_slot + (size_t)(buf ^ 1u) * KC;
    const u32 M = KE + D, m0 = b * KW_CH, m1 = m0 + KW_CH < M ? m0 + KW_CH : M, cm = m1 - m0;   // (b < ceil(M / KW_CH): the caller saw to it)
    // the kept and the delta CSR hold COMPACT node ids (sg_kept_compact: a warm window has no raw outbound IP); the window's arrays dense ones
    const u32 MK = d.max_known, NKn = (u32)d.ctr[C_N_KNOWN], NC = d.max_known + d.max_labels;
    auto dn = [&](u32 c) -> u32 { return c < MK ? c : NKn + (c - MK); };
    auto has_dense = [&](u32 c) -> bool { return c < NKn || c >= MK; };   // (compact rows [N_KNOWN, max_known): ids no node has yet)
    const bool lastc = m1 == M;
    if (t < 64) { const u32 r = kw_merge_path(kfrom, kcol, d.dc_from, d.dc_col, KE, D, m0); if (t == 0) dg[0] = r; }
    else if (t < 128) { const u32 r = kw_merge_path(kfrom, kcol, d.dc_from, d.dc_col, KE, D, m1); if (t == 64) dg[1] = r; }
    for (u32 i = t; i < KW_ROWS * 5; i += KW_THREADS) kw_racc[i] = 0;
    if (t < KW_CH / 32) tbits[t] = 0;
    __syncthreads();
    const u32 i0 = dg[0], i1 = dg[1], j0 = m0 - i0, j1 = m1 - i1, na = i1 - i0, nd = j1 - j0;   // na + nd = cm
    // element s of the chunk's concatenated list: kept edge i0 + s (s < na) or new edge j0 + s - na
    u32 fr[KW_Q], co[KW_Q], sl[KW_Q]; ulonglong2 x[KW_Q], y[KW_Q]; bool have[KW_Q], tc[KW_Q];
#pragma unroll
    for (int q = 0; q < KW_Q; q++) {
        const u32 s = (u32)q * KW_THREADS + t;
        have[q] = s < cm;
        if (!(have[q] && s >= na)) {
            const u32 ic = (have[q] && s < na) ? i0 + s : 0u;        // (a thread without an element: kept position 0 — KE >= 1 on a warm window — ignored)
            fr[q] = kfrom[ic]; co[q] = kcol[ic]; sl[q] = kslot[ic];
            const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.k_acc + (size_t)ic * 4);
            x[q] = a[0]; y[q] = a[1];
        } else {
            const u32 j = j0 + (s - na);
            fr[q] = d.dc_from[j]; co[q] = d.dc_col[j]; sl[q] = d.dl_img[d.dc_slot[j]];
            const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.dc_acc + (size_t)j * 4);
            x[q] = a[0]; y[q] = a[1];
        }
        if (have[q]) { kF[s] = fr[q]; kC[s] = co[q]; }
        tc[q] = have[q] && (y[q].x >> 63) != 0;                      // (a new edge is touched by construction: pass B set the bit)
    }
    __syncthreads();
    SG_STAMP(d, 2, 1);
    // merged index: own index in its list + the elements of the OTHER list below its key
    u32 ml[KW_Q];
#pragma unroll
    for (int q = 0; q < KW_Q; q++) {
        const u32 s = (u32)q * KW_THREADS + t;
        ml[q] = 0;
        if (!have[q]) continue;
        const bool kept = s < na;
        u32 lo = kept ? na : 0u, hi = kept ? cm : na;
        while (lo < hi) { const u32 m = (lo + hi) >> 1; if (kw_key_less(kF[m], kC[m], fr[q], co[q])) lo = m + 1; else hi = m; }
        ml[q] = kept ? s + (lo - na) : (s - na) + lo;
        if (tc[q]) atomicOr(&tbits[ml[q] >> 5], 1u << (ml[q] & 31u));
    }
    __syncthreads();
    if (t < 64) {                                                    // one wave: exclusive prefix over the 64 words' popcounts
        const u32 c = (u32)__popc(tbits[t]);
        u32 incl = c;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const u32 o = __shfl_up(incl, s2, 64); if ((int)lane >= s2) incl += o; }
        tpre[t] = incl - c;
        if (t == 63) {
            tpre[64] = incl;
            __hip_atomic_store(&d.kw_tot[b], ((u64)epoch << 32) | incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *pre = 0;
        }
    }
    // rows of the chunk's first and last element, of the element before the chunk (all uniform; through LDS)
    if (t == 128) {
        // the merged element m0 - 1 / m1 - 1 is the larger of the last kept and the last new key before the cut
        auto row_before = [&](u32 i, u32 j) -> u32 {                 // row of the last of the first i kept + j new keys (i + j >= 1)
            if (!j) return kfrom[i - 1];
            if (!i) return d.dc_from[j - 1];
            const u32 a = kfrom[i - 1], c = d.dc_from[j - 1];
            return a > c ? a : c;                                    // (keys ascend in both lists: the later row is the later key's)
        };
        dg[2] = m0 ? row_before(i0, j0) + 1u : 0u;                   // v_lo: rows that START in this chunk begin behind the row of element m0 - 1
        dg[3] = row_before(i1, j1);                                  // v_hi: the row of the chunk's last element
    }
    __syncthreads();
    {   // look-back: touched elements of the chunks before this one
        u32 mine = 0;
        for (u32 j = t; j < b; j += KW_THREADS) {
            u64 w;
            do { w = __hip_atomic_load(&d.kw_tot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((u32)(w >> 32) != epoch);
            mine += (u32)w;
        }
        if (b) { mine = wave_sum_u32(mine); if (lane == 0 && mine) atomicAdd(pre, mine); }
    }
    __syncthreads();
    SG_STAMP(d, 2, 2);
    const u32 base = *pre, total = tpre[64];
    const u32 ME = (u32)d.max_edges;
    const u32 v_lo = dg[2], v_hi = dg[3];
    const u32 va = na ? kF[0] : 0xFFFFFFFFu, vb = nd ? kF[na] : 0xFFFFFFFFu, v0 = va < vb ? va : vb;   // the chunk's first row: the smaller of the two lists' first rows
    auto rank_excl = [&](u32 mi) -> u32 {                            // touched elements of the chunk below merged index mi (mi <= cm)
        if (mi >= KW_CH) return total;
        return tpre[mi >> 5] + (u32)__popc(tbits[mi >> 5] & ((1u << (mi & 31u)) - 1u));
    };
    auto fold = [&](u32 row, u64 cnt_, u64 err, u64 sum, u64 ssq, u64 mx) {
        const u32 r = row - v0;
        if (r < KW_ROWS) {
            u64* a = kw_racc + (size_t)r * 5;
            if (cnt_) atomicAdd(&a[0], cnt_);
            if (err) atomicAdd(&a[1], err);
            if (sum) atomicAdd(&a[2], sum);
            if (ssq) atomicAdd(&a[3], ssq);
            if (mx) atomicMax(&a[4], mx);
        } else {
            u64* g = d.st_sum + (size_t)dn(row) * SG_NODE_STAT_SUM_WORDS;
            if (cnt_) atomicAdd(&g[ST_OUT_CNT], cnt_);
            if (err) atomicAdd(&g[ST_OUT_ERR], err);
            if (sum) atomicAdd(&g[ST_OUT_SUM], sum);
            if (ssq) atomicAdd(&g[ST_OUT_SSQ], ssq);
            if (mx) atomicMax(&d.st_max[(size_t)dn(row) * 2], mx);
        }
    };
    bool wsame[KW_Q], inw[KW_Q];
#pragma unroll
    for (int q = 0; q < KW_Q; q++) {
        const u32 f0 = rdlane32(fr[q], 0);
        wsame[q] = __ballot((have[q] && fr[q] == f0) ? 1 : 0) == ~0ull && f0 - v0 < KW_ROWS;
    }
#pragma unroll
    for (int q = 0; q < KW_Q; q++) {
        inw[q] = false;
        if (!have[q]) continue;
        const u32 nk = m0 + ml[q];                                   // the element's position in the new kept CSR
        ncol[nk] = co[q]; nfrom[nk] = fr[q]; nslot[nk] = sl[q];
        d.wk_pos[sl[q]] = nk;
        if (!tc[q]) continue;
        const u64 np = (u64)base + rank_excl(ml[q]);
        if (np >= ME) continue;
        inw[q] = true;
        const u64 mx = y[q].x & ~(1ull << 63);
        d.col[np] = dn(co[q]); d.csr_from[np] = dn(fr[q]); d.alive_csr[np] = 0;
        ulonglong2* o = reinterpret_cast<ulonglong2*>(d.acc_csr + (size_t)np * 4);
        o[0] = x[q]; o[1] = make_ulonglong2(mx, y[q].y);
        if (!wsame[q]) fold(fr[q], x[q].x & 0xFFFFFFFFull, x[q].x >> 32, x[q].y, y[q].y, mx);
    }
#pragma unroll
    for (int q = 0; q < KW_Q; q++) if (wsame[q]) {                   // (uniform per wave) the wave's 64 elements lie in one row: reduced in the wave
        const bool in = inw[q];
        const u64 c_ = wave_sum_u64(in ? x[q].x & 0xFFFFFFFFull : 0ull), e_ = wave_sum_u64(in ? x[q].x >> 32 : 0ull);
        const u64 s_ = wave_sum_u64(in ? x[q].y : 0ull), q_ = wave_sum_u64(in ? y[q].y : 0ull), m_ = wave_max_u64(in ? y[q].x & ~(1ull << 63) : 0ull);
        if (lane == 0) fold(rdlane32(fr[q], 0), c_, e_, s_, q_, m_);
    }
    SG_STAMP(d, 2, 3);
    // row pointers of the rows that start in this chunk: a row starts at the merged position krp + dc_rowptr
    for (u32 v = v_lo + t; v <= v_hi; v += KW_THREADS) {
        const u32 ns = krp[v] + d.dc_rowptr[v];
        nrp[v] = ns;
        const u64 rp = (u64)base + rank_excl(ns - m0);
        if (has_dense(v)) d.rowptr[dn(v)] = rp < ME ? (u32)rp : ME;
    }
    if (lastc) {
        const u64 Ef = (u64)base + total;
        for (u32 v = v_hi + 1 + t; v <= NC; v += KW_THREADS) { nrp[v] = M; if (v < NC && has_dense(v)) d.rowptr[dn(v)] = Ef < ME ? (u32)Ef : ME; }
        if (t == 0) d.rowptr[N] = Ef < ME ? (u32)Ef : ME;
        if (t == 0) { d.ctr[C_N_EDGES] = Ef < ME ? Ef : ME; d.ctr[C_EDGES_FOUND] = Ef; if (Ef > ME) d.ctr[C_DROPPED_CAP] += Ef - ME; }
    }
    __syncthreads();
    SG_STAMP(d, 2, 4);
    {
        const u32 nr = v_hi - v0 + 1 < KW_ROWS ? v_hi - v0 + 1 : KW_ROWS;
        for (u32 r = t; r < nr; r += KW_THREADS) {
            const u64* a = kw_racc + (size_t)r * 5;
            const u64 cnt_ = a[0], err = a[1], sum = a[2], ssq = a[3], mx = a[4];
            if (!(cnt_ | err | sum | ssq | mx)) continue;
            const u32 v = dn(v0 + r);
            u64* g = d.st_sum + (size_t)v * SG_NODE_STAT_SUM_WORDS;
            if (cnt_) atomicAdd(&g[ST_OUT_CNT], cnt_);
            if (err) atomicAdd(&g[ST_OUT_ERR], err);
            if (sum) atomicAdd(&g[ST_OUT_SUM], sum);
            if (ssq) atomicAdd(&g[ST_OUT_SSQ], ssq);
            if (mx) atomicMax(&d.st_max[(size_t)v * 2], mx);
        }
    }
}
__global__ __launch_bounds__(KW_THREADS) void kw_compact(Dev d, u32 epoch, u64* scratch_sum, u64* scratch_max, u64 seq, u32 shared_chip) {
    extern __shared__ u64 kw_racc[];                                 // [KW_ROWS][5]: cnt, err, sum, ssq, max
    __shared__ u64 bal[KW_Q][KW_NW];
    __shared__ u32 wpre[KW_Q][KW_NW];
    __shared__ u32 qpre[KW_Q + 1];
    __shared__ u32 pre, bdyn;
    if (blockIdx.x < KW_CAPW) { kw_capture(d, scratch_sum, scratch_max, blockIdx.x, KW_CAPW, KW_THREADS); return; }
    const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const u32 KE = (u32)d.ctr[C_KEPT_E], N = (u32)d.ctr[C_N_NODES];
    const u32 buf = (u32)d.ctr[C_KEPT_BUF] & 1u;                      // the current kept buffer (a delta window writes the other one and k3_in_part flips)
    const u32 D = (!d.ctr[C_COLD] && d.ctr[C_DELTA_N]) ? d.dc_rowptr[d.max_known + d.max_labels] : 0u;   // (uniform) new edges of a warm window, sorted by the delta chain (compact rows)
    const u32 nchunk = (KE + D) ? (KE + D + KW_CH - 1) / KW_CH : 1u; // (a delta window's chunks cut the MERGE of the kept and the new edges: kw_compact_delta)
    const u32 G = gridDim.x - KW_CAPW;                               // chunk workgroups of the launch
    u32 b = blockIdx.x - KW_CAPW;
    // Order by ticket (see k2_rowptr) only when the chunks that DO something cannot all be resident at once — three workgroups per CU.
    // The grid is sized for the kept arrays' capacity; the chunks behind the last kept edge
    // return at once and free their place, so up to KW_RESIDENT working chunks never wait for one that cannot start,
    // whatever the dispatch order.  (586 same-address ticket draws were ~7 us at the head of every launch.)
    // (shared_chip — an engine with several windows in flight: another slot's look-back kernel may hold CUs at the same time, and two launches
    // whose resident chunks each wait for a chunk that cannot start would wait for ever; by ticket a chunk only ever waits for chunks that
    // have started — ADVICE r5)
    if (nchunk > KW_RESIDENT || shared_chip) {                       // (uniform: every workgroup reads the same count)
        if (t == 0) { const u32 tk = atomicAdd(&d.lb_ticket[1], 1u); if (tk == G - 1) atomicExch(&d.lb_ticket[1], 0u); bdyn = tk; }
        __syncthreads();
        b = bdyn;
    }
    if (b >= nchunk) return;                                         // (nobody waits for a chunk behind its own)
    SG_STAMP(d, 2, 0);
    if (b == 0 && t == 0) {
        d.ctr[C_OVF_N] = 0;                                          // pass B has consumed the overflow list
        d.ctr[C_ACT_L] = SG_ACT_NONE; d.ctr[C_ACT_P] = 0;
        d.ctr[C_HUB_ITEMS] = 0;                                      // the hub blocks of the WINDOW's rows are listed behind this kernel (kw_finish_rows); a rebuild's were the kept rows'
        if (!d.ctr[C_COLD]) d.ctr[C_WARM_WINDOWS] += 1;
        if (D) d.ctr[C_DELTA_WINDOWS] += 1;
        // for the host's policy (it never waits for the device: it reads this note, a window or two late, when it closes a later window)
        d.host_note[1] = d.ctr[C_COLD] | (d.ctr[C_N_OBIP] ? 0x100ull : 0ull);
        __threadfence_system();
        d.host_note[0] = seq;
    }
    if (KE == 0) {                                                   // an empty kept set: an empty window
        for (u32 v = t; v <= N; v += KW_THREADS) d.rowptr[v] = 0;
        if (t == 0) { d.ctr[C_N_EDGES] = 0; d.ctr[C_EDGES_FOUND] = 0; }
        return;
    }
    if (D) { kw_compact_delta(d, b, epoch, KE, N, D, buf, kw_racc, &pre); SG_STAMP(d, 2, 5); return; }
    const u32* __restrict__ kcol = buf ? d.k_col2 : d.k_col; const u32* __restrict__ kfrom = buf ? d.k_from2 : d.k_from; const u32* __restrict__ krp = buf ? d.k_rowptr2 : d.k_rowptr;
    // The kept CSR holds COMPACT node ids unless this window has raw outbound IPs (sg_kept_compact: then the rebuild has just written it in
    // dense ids and the state is invalid anyway); the window's arrays hold dense ids: known ids as they are, labels from N_KNOWN on.
    const bool cmp = d.ctr[C_N_OBIP] == 0;
    const u32 MK = d.max_known, NKn = (u32)d.ctr[C_N_KNOWN], NR = cmp ? d.max_known + d.max_labels : N;   // rows of the kept CSR
    auto dn = [&](u32 c) -> u32 { return (!cmp || c < MK) ? c : NKn + (c - MK); };
    auto has_dense = [&](u32 c) -> bool { return !cmp || c < NKn || c >= MK; };
    const u32 p0 = b * KW_CH, last = (p0 + KW_CH < KE ? p0 + KW_CH : KE) - 1;
    u32 fr[KW_Q], co[KW_Q]; ulonglong2 x[KW_Q], y[KW_Q]; bool tc[KW_Q];
#pragma unroll
    for (int q = 0; q < KW_Q; q++) {
        const u32 i = p0 + (u32)q * KW_THREADS + t, ic = i <= last ? i : last;
        fr[q] = kfrom[ic]; co[q] = kcol[ic];
        const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.k_acc + (size_t)ic * 4);
        x[q] = a[0]; y[q] = a[1];
    }
    const u32 v0 = kfrom[p0], v_hi = kfrom[last];                    // first and last row with an edge in this chunk
    const u32 v_lo = b == 0 ? 0u : kfrom[p0 - 1] + 1u;               // rows that START here: (row of the position before the chunk, v_hi]
    for (u32 i = t; i < KW_ROWS * 5; i += KW_THREADS) kw_racc[i] = 0;
#pragma unroll
    for (int q = 0; q < KW_Q; q++) {
        tc[q] = p0 + (u32)q * KW_THREADS + t <= last && (y[q].x >> 63) != 0;
        const u64 m = __ballot(tc[q] ? 1 : 0);
        if (lane == 0) { bal[q][wave] = m; wpre[q][wave] = (u32)__popcll(m); }
    }
    __syncthreads();
    SG_STAMP(d, 2, 1);
    if (t < KW_Q) { u32 acc = 0; for (u32 w2 = 0; w2 < KW_NW; w2++) { const u32 c = wpre[t][w2]; wpre[t][w2] = acc; acc += c; } qpre[t + 1] = acc; }
    __syncthreads();
    if (t == 0) {
        u32 run = 0;
        for (int q = 0; q < KW_Q; q++) { const u32 c = qpre[q + 1]; qpre[q] = run; run += c; }
        qpre[KW_Q] = run;
        __hip_atomic_store(&d.kw_tot[b], ((u64)epoch << 32) | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (relaxed: see k2_rowptr)
        pre = 0;
    }
    __syncthreads();
    {
        u32 mine = 0;
        for (u32 j = t; j < b; j += KW_THREADS) {
            u64 w;
            do { w = __hip_atomic_load(&d.kw_tot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((u32)(w >> 32) != epoch);
            mine += (u32)w;
        }
        if (b) { mine = wave_sum_u32(mine); if (lane == 0 && mine) atomicAdd(&pre, mine); }
    }
    __syncthreads();
    SG_STAMP(d, 2, 2);
    const u32 base = pre, total = qpre[KW_Q];
    const u64 lt = (1ull << lane) - 1ull;
    // (row statistics: 64 lanes adding to the same five LDS words serialise in the LDS unit — a wave whose positions all lie in one row
    // sums in registers first and sends one set of atomics)
    bool wsame[KW_Q];
#pragma unroll
    for (int q = 0; q < KW_Q; q++) {
        const u32 f0 = rdlane32(fr[q], 0);
        wsame[q] = __ballot(fr[q] == f0 ? 1 : 0) == ~0ull && f0 - v0 < KW_ROWS;
    }
    const u32 ME = (u32)d.max_edges;                                 // (the kept arrays hold npb x pcap edges; a WINDOW's rows stop at the configured capacity: cut and counted, as k2_rowptr does)
#pragma unroll
    for (int q = 0; q < KW_Q; q++) if (tc[q]) {
        const u32 np = base + qpre[q] + wpre[q][wave] + (u32)__popcll(bal[q][wave] & lt);
        if (np >= ME) continue;
        const u64 mx = y[q].x & ~(1ull << 63);
        d.col[np] = dn(co[q]); d.csr_from[np] = dn(fr[q]); d.alive_csr[np] = 0;
        ulonglong2* o = reinterpret_cast<ulonglong2*>(d.acc_csr + (size_t)np * 4);
        o[0] = x[q]; o[1] = make_ulonglong2(mx, y[q].y);
        const u64 cnt = x[q].x & 0xFFFFFFFFull, err = x[q].x >> 32;
        const u32 r = fr[q] - v0;
        if (r < KW_ROWS && wsame[q]) {                               // the wave's 64 positions lie in ONE row (hub rows: half of C3's edges): reduced in the wave below
        } else if (r < KW_ROWS) {
            u64* a = kw_racc + (size_t)r * 5;
            if (cnt) atomicAdd(&a[0], cnt);
            if (err) atomicAdd(&a[1], err);
            if (x[q].y) atomicAdd(&a[2], x[q].y);
            if (y[q].y) atomicAdd(&a[3], y[q].y);
            if (mx) atomicMax(&a[4], mx);
        } else {
            u64* g = d.st_sum + (size_t)dn(fr[q]) * SG_NODE_STAT_SUM_WORDS;
            if (cnt) atomicAdd(&g[ST_OUT_CNT], cnt);
            if (err) atomicAdd(&g[ST_OUT_ERR], err);
            if (x[q].y) atomicAdd(&g[ST_OUT_SUM], x[q].y);
            if (y[q].y) atomicAdd(&g[ST_OUT_SSQ], y[q].y);
            if (mx) atomicMax(&d.st_max[(size_t)dn(fr[q]) * 2], mx);
        }
    }
#pragma unroll
    for (int q = 0; q < KW_Q; q++) if (wsame[q]) {                   // (uniform per wave)
        const bool in = tc[q] && base + qpre[q] + wpre[q][wave] + (u32)__popcll(bal[q][wave] & lt) < ME;
        const u64 cnt = wave_sum_u64(in ? x[q].x & 0xFFFFFFFFull : 0ull), err = wave_sum_u64(in ? x[q].x >> 32 : 0ull);
        const u64 sum = wave_sum_u64(in ? x[q].y : 0ull), ssq = wave_sum_u64(in ? y[q].y : 0ull), mx = wave_max_u64(in ? y[q].x & ~(1ull << 63) : 0ull);
        if (lane == 0) {
            u64* a = kw_racc + (size_t)(rdlane32(fr[q], 0) - v0) * 5;
            if (cnt) atomicAdd(&a[0], cnt);
            if (err) atomicAdd(&a[1], err);
            if (sum) atomicAdd(&a[2], sum);
            if (ssq) atomicAdd(&a[3], ssq);
            if (mx) atomicMax(&a[4], mx);
        }
    }
    SG_STAMP(d, 2, 3);
    // new row pointers of the rows that start in this chunk: rank of the row's first kept position among the chunk's touched ones
    for (u32 v = v_lo + t; v <= v_hi; v += KW_THREADS) {
        const u32 xl = krp[v] - p0, q = xl / KW_THREADS, tt = xl % KW_THREADS, w2 = tt >> 6, l2 = tt & 63u;
        const u32 rp = base + qpre[q] + wpre[q][w2] + (u32)__popcll(bal[q][w2] & ((1ull << l2) - 1ull));
        if (has_dense(v)) d.rowptr[dn(v)] = rp < ME ? rp : ME;
    }
    if (b == nchunk - 1) {                                           // the last chunk knows E; the rows behind the last kept edge are empty
        const u32 Ef = base + total, E = Ef < ME ? Ef : ME;
        for (u32 v = v_hi + 1 + t; v < NR; v += KW_THREADS) if (has_dense(v)) d.rowptr[dn(v)] = E;
        if (t == 0) d.rowptr[N] = E;
        if (t == 0) { d.ctr[C_N_EDGES] = E; d.ctr[C_EDGES_FOUND] = Ef; if (Ef > ME) d.ctr[C_DROPPED_CAP] += (u64)(Ef - ME); }
    }
    __syncthreads();                                                 // every LDS fold is in
    SG_STAMP(d, 2, 4);
    {
        const u32 nr = v_hi - v0 + 1 < KW_ROWS ? v_hi - v0 + 1 : KW_ROWS;
        for (u32 r = t; r < nr; r += KW_THREADS) {
            const u64* a = kw_racc + (size_t)r * 5;
            const u64 cnt = a[0], err = a[1], sum = a[2], ssq = a[3], mx = a[4];
            if (!(cnt | err | sum | ssq | mx)) continue;
            const u32 v = v0 + r, vd = dn(v);
            u64* g = d.st_sum + (size_t)vd * SG_NODE_STAT_SUM_WORDS;
            if (v >= v_lo && v < v_hi) {                             // wholly inside this chunk: nobody else writes the row (the arrays were zeroed by the window reset)
                g[ST_OUT_CNT] = cnt; g[ST_OUT_ERR] = err; g[ST_OUT_SUM] = sum; g[ST_OUT_SSQ] = ssq; d.st_max[(size_t)vd * 2] = mx;
            } else {
                if (cnt) atomicAdd(&g[ST_OUT_CNT], cnt);
                if (err) atomicAdd(&g[ST_OUT_ERR], err);
                if (sum) atomicAdd(&g[ST_OUT_SUM], sum);
                if (ssq) atomicAdd(&g[ST_OUT_SSQ], ssq);
                if (mx) atomicMax(&d.st_max[(size_t)vd * 2], mx);
            }
        }
    }
    SG_STAMP(d, 2, 5);
}
// behind kw_compact (run by extra workgroups of k3_in_reduce on a warm window): a thread per node — out-degree from the new row pointers,
// mean / deviation of the row's out-events from its sums (the row sort's own expressions), the hub rows' block work items
__device__ __forceinline__ void kw_finish_rows(const Dev& d, u32 tid, u32 nt) {
    const u32 N = (u32)d.ctr[C_N_NODES];
    for (u32 v = tid; v < N; v += nt) {
        const u32 s0 = d.rowptr[v], dg = d.rowptr[v + 1] - s0;
        u64* t = d.st_sum + (size_t)v * SG_NODE_STAT_SUM_WORDS;
        const u64 tc = t[ST_OUT_CNT], ts = t[ST_OUT_SUM], tq = t[ST_OUT_SSQ];
        t[ST_OUT_DEG] = dg;
        d.row_mu[v] = mean_us(ts, tc); d.row_sd[v] = std_us(ts, tq, tc);
        if (dg > SG_MEAN_BLOCK) {
            const u32 nblk = (dg + SG_MEAN_BLOCK - 1) / SG_MEAN_BLOCK;
            const u32 ib = (u32)atomicAdd(&d.ctr[C_HUB_ITEMS], (u64)nblk);   // (zeroed by kw_compact)
            d.hub_base[v] = ib;
            for (u32 j = 0; j < nblk; j++) if (ib + j < d.hub_cap) d.hub_items[ib + j] = make_uint2(v, j);
        }
    }
}

// ---- in-statistics: per destination node, reduce over its in-edges --------------------------------
// No device-scope atomics (they top out at ~22 G/s chip-wide: the hashed-LDS + atomic-flush version of round 1 spent
// 175 us on C3's 1 M edges = 2.4 % of the HBM roofline).  Two launches instead:
//   k3_in_part   grid = node ranges x edge slices.  Workgroup (r, s) owns the K3_IN_NR nodes of range r in node-indexed
//                LDS arrays, scans slice s of the CSR destination column (coalesced u32 reads, L2-resident across the
//                ranges) and folds the accumulators of the edges that point into its range with LDS atomics; then it
//                writes its arrays to the partial buffer with plain coalesced stores.
//   k3_in_reduce one thread per (node, word): sums (max for the last word) the slices' partials into st_sum / st_max.
// Exact (integer sums and max are order-free) and deterministic.
#define K3_IN_NR    3072      // nodes per range: 3072 x 6 x 8 B = 144 KiB of LDS
#define K3_IN_SMAX  32        // edge slices at most
// The window's open connections (SG_EV_ALIVE, f-2) are marked here too: every record's edge exists in
// the CSR (K1 created it with count 0 if it carried no request); a binary search in the sorted row
// finds it.  Costs one scalar load when the window has none.
__device__ __forceinline__ void alive_mark(const Dev& d, u32 g, u32 G, u32 t) {
    const u64 n_all = d.ctr[C_ALIVE_N];
    if (n_all == 0) return;
    const u32 n = (u32)(n_all < d.alive_cap ? n_all : d.alive_cap);
    const u32 nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS], nob = (u32)d.ctr[C_N_OBIP];
    for (u32 i = g * 1024 + t; i < n; i += G * 1024) {
        const u64 key = d.alive_keys[i];
        const u32 f = dense_of(d, (u32)(key >> 32), nk, nl, nob), to = dense_of(d, (u32)key, nk, nl, nob);
        bool ok = f != SG_NONE && to != SG_NONE;
        if (ok) {
            u32 lo = d.rowptr[f], hi = d.rowptr[f + 1];
            if ((u64)hi > d.max_edges) hi = (u32)d.max_edges;
            const u32 end = hi;
            while (lo < hi) { const u32 m = (lo + hi) >> 1; if (d.col[m] < to) lo = m + 1; else hi = m; }
            ok = lo < end && d.col[lo] == to;
            if (ok) {
                atomicAdd(&d.alive_csr[lo], 1u);
                atomicAdd(&d.st_sum[(size_t)f * SG_NODE_STAT_SUM_WORDS + ST_OUT_ALIVE], 1ull);
                atomicAdd(&d.st_sum[(size_t)to * SG_NODE_STAT_SUM_WORDS + ST_IN_ALIVE], 1ull);
            }
        }
        if (!ok) atomicAdd(&d.ctr[C_ALIVE_DROP], 1ull);
    }
}

__global__ __launch_bounds__(1024) void k3_in_part(Dev d, u32 S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 E = (u32)d.ctr[C_N_EDGES], N = (u32)d.ctr[C_N_NODES];
    const u32 G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
    SG_STAMP(d, 3, 0);
    // a delta window's kw_compact (the launch before this one) has written the kept CSR, grown by the window's new edges, to the other
    // buffer: flip (nothing in this launch reads the kept state; the next window's pass B and kw_compact do)
    if (g == 0 && t == 0 && d.warm && !d.ctr[C_COLD] && d.ctr[C_DELTA_N]) {
        d.ctr[C_KEPT_E] += (u64)d.dc_rowptr[d.max_known + d.max_labels];
        d.ctr[C_KEPT_BUF] ^= 1ull;                                    // (C_DELTA_N stays for the window's reader: sg_stats.windows_delta; kc_prepare re-arms it)
    }
    alive_mark(d, g, G, t);
    const u32 r = g / S, sl = g % S, n0 = r * K3_IN_NR;
    if (n0 >= N) return;
    const u32 nr = N - n0 < K3_IN_NR ? N - n0 : K3_IN_NR;
    u64* acc = reinterpret_cast<u64*>(smem);                         // [nr][6]: deg, cnt, err, sum, ssq, max
    for (u32 i = t; i < nr * 6; i += 1024) acc[i] = 0;
    __syncthreads();
    SG_STAMP(d, 3, 1);
    const u32 per = (E + S - 1) / S, p0 = sl * per < E ? sl * per : E, p1 = p0 + per < E ? p0 + per : E;
    // Eight edges per thread and trip; the destinations of the NEXT trip are fetched behind this trip's accumulator loads, so a
    // trip costs one round trip, not two (a slice of C3 is 31 k edges: four trips; at four edges per trip and no lookahead the
    // sixteen dependent round trips were most of this kernel's 22 us).
    constexpr int K3Q = 8;
    u32 nxt[K3Q];
#pragma unroll
    for (int q = 0; q < K3Q; q++) { const u32 p = p0 + t + q * 1024; nxt[q] = p < p1 ? d.col[p] - n0 : 0xFFFFFFFFu; }
    for (u32 pb = p0 + t; pb < p1; pb += 1024 * K3Q) {
        u32 to[K3Q];
        ulonglong2 x[K3Q], y[K3Q];
#pragma unroll
        for (int q = 0; q < K3Q; q++) {
            to[q] = nxt[q];
            if (to[q] < nr) { const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)(pb + q * 1024) * 4); x[q] = a[0]; y[q] = a[1]; }
        }
#pragma unroll
        for (int q = 0; q < K3Q; q++) { const u32 p = pb + 1024 * K3Q + q * 1024; nxt[q] = p < p1 ? d.col[p] - n0 : 0xFFFFFFFFu; }
#pragma unroll
        for (int q = 0; q < K3Q; q++) if (to[q] < nr) {
            u64* o = acc + (size_t)to[q] * 6;
            atomicAdd(&o[0], 1ull); atomicAdd(&o[1], x[q].x & 0xFFFFFFFFull); if (x[q].x >> 32) atomicAdd(&o[2], x[q].x >> 32);
            atomicAdd(&o[3], x[q].y); atomicAdd(&o[4], y[q].y); atomicMax(&o[5], y[q].x);
        }
    }
    SG_STAMP(d, 3, 2);
    __syncthreads();
    SG_STAMP(d, 3, 3);
    u64* out = d.in_part + ((size_t)r * S + sl) * K3_IN_NR * 6;
    for (u32 i = t; i < nr * 6; i += 1024) out[i] = acc[i];
    SG_STAMP(d, 3, 4);
}
__global__ __launch_bounds__(256) void k3_in_reduce(Dev d, u32 S, u32 fin) {
    const u32 N = (u32)d.ctr[C_N_NODES];
    // the out-statistics of the rows the row sort took block by block: the launch's LAST workgroup, and nothing else there (three
    // dependent round trips — in front of the reduction they were on the path of the threads that ran both)
    // (an engine that keeps state: the last `fin` workgroups finish EVERY row behind kw_compact instead — degree, mean / deviation, hub work items)
    if (blockIdx.x >= gridDim.x - fin) {
        if (d.warm) kw_finish_rows(d, (blockIdx.x - (gridDim.x - fin)) * 256 + threadIdx.x, fin * 256);
        else if (blockIdx.x == gridDim.x - 1) k2_split_finish(d, threadIdx.x, 256);
        return;
    }
    const u32 GW = gridDim.x - fin;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < (u64)N * 6; i += (u64)GW * 256) {
        const u32 v = (u32)(i / 6), k = (u32)(i % 6), r = v / K3_IN_NR;
        const u64* p = d.in_part + ((size_t)r * S * K3_IN_NR + (v - r * K3_IN_NR)) * 6 + k;
        const size_t st = (size_t)K3_IN_NR * 6;
        u64 a = 0;
        u32 sl = 0;
        // (measured: all 32 slices in one batch of loads instead of four batches of eight — 10.1 -> 12.1 us on one kind of box)
        if (k == 5) {
            for (; sl + 8 <= S; sl += 8) {                           // eight independent loads in flight
                u64 x[8];
#pragma unroll
                for (int q = 0; q < 8; q++) x[q] = p[(size_t)(sl + q) * st];
#pragma unroll
                for (int q = 0; q < 8; q++) a = x[q] > a ? x[q] : a;
            }
            for (; sl < S; sl++) { const u64 x = p[(size_t)sl * st]; a = x > a ? x : a; }
            d.st_max[(size_t)v * 2 + 1] = a;
        } else {
            for (; sl + 8 <= S; sl += 8) {
                u64 x[8];
#pragma unroll
                for (int q = 0; q < 8; q++) x[q] = p[(size_t)(sl + q) * st];
#pragma unroll
                for (int q = 0; q < 8; q++) a += x[q];
            }
            for (; sl < S; sl++) a += p[(size_t)sl * st];
            d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + (k == 0 ? ST_IN_DEG : k == 1 ? ST_IN_CNT : k == 2 ? ST_IN_ERR : k == 3 ? ST_IN_SUM : ST_IN_SSQ)] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3  node_features: fp32 x_v from the integer node statistics.
// ------------------------------------------------------------------------------------------------
// Two lanes per node: lane 0 of the pair turns the out-side statistics into features, lane 1 the
// in-side ones (the fp64 log1p / sqrt chains are the whole cost of this kernel), then they swap.
// Workgroups [0, nb_nodes) do the nodes; the rest do the edge features (one thread per edge, edge_features()).
__global__ __launch_bounds__(256) void k3_node_features(Dev d, u32 nb_nodes) {
    if (blockIdx.x >= nb_nodes) {
        const u32 E = (u32)d.ctr[C_N_EDGES];
        for (u32 p = (blockIdx.x - nb_nodes) * 256 + threadIdx.x; p < E; p += (gridDim.x - nb_nodes) * 256) edge_features(d, p);
        return;
    }
    const u32 N = (u32)d.ctr[C_N_NODES], nk = (u32)d.ctr[C_N_KNOWN];
    const u32 side = threadIdx.x & 1u;
    for (u32 v0 = blockIdx.x * 128; v0 < N; v0 += nb_nodes * 128) {
        const u32 v = v0 + (threadIdx.x >> 1);
        const bool live = v < N;
        float a[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (live) {
            const u64* s = d.st_sum + (size_t)v * SG_NODE_STAT_SUM_WORDS;
            const u64 dg = s[ST_OUT_DEG + side], c = s[ST_OUT_CNT + side], er = s[ST_OUT_ERR + side], sm = s[ST_OUT_SUM + side], sq = s[ST_OUT_SSQ + side];
            const u64 mx = d.st_max[(size_t)v * 2 + side];
            a[0] = (float)log1p((double)dg);
            a[1] = (float)log1p((double)c);
            a[2] = (float)log1p(mean_us(sm, c) / 1000.0);
            a[3] = c ? (float)((double)er / (double)c) : 0.0f;
            a[4] = (float)log1p((double)mx / 1e6);
            a[5] = (float)log1p(std_us(sm, sq, c) / 1000.0);
            a[6] = (float)log1p((double)s[ST_OUT_ALIVE + side]);
        }
        float b[7];
#pragma unroll
        for (int k = 0; k < 7; k++) b[k] = __shfl_xor(a[k], 1, 64);
        if (!live) continue;
        float4* o = reinterpret_cast<float4*>(d.x0 + (size_t)v * SG_F_IN);
        if (side == 0) {                                             // a = out side, b = in side
            const u32 kind = v < nk ? d.kind[v] : 0u;
            o[0] = make_float4(a[0], b[0], a[1], b[1]);
            o[1] = make_float4(a[2], b[2], a[3], b[3]);
            o[2] = make_float4(a[4], b[4], kind == SG_NODE_POD ? 1.0f : 0.0f, kind == SG_NODE_SERVICE ? 1.0f : 0.0f);
            o[3] = make_float4(kind == 0 ? 1.0f : 0.0f, a[5], b[5], 1.0f);
        } else {                                                     // a = in side, b = out side
            o[4] = make_float4(b[6], a[6], 0.0f, 0.0f);
#pragma unroll
            for (int q = 5; q < (int)SG_F_IN / 4; q++) o[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {                       // the alive list is consumed (k3_in_stats): report and re-arm
        const u64 n = d.ctr[C_ALIVE_N];
        d.ctr[C_ALIVE_SEEN] = n;
        d.ctr[C_ALIVE_DROPPED] = d.ctr[C_ALIVE_DROP] + (n > d.alive_cap ? n - d.alive_cap : 0);
        d.ctr[C_ALIVE_N] = 0; d.ctr[C_ALIVE_DROP] = 0;
    }
}

// window reset in one launch (replaces seven memsets): node arrays, outbound-ip table, window
// counters; for variant 1 also the whole edge table if more edges were found than max_edges.
__global__ __launch_bounds__(256) void k_reset_window(Dev d) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, nt = (u64)gridDim.x * 256;
    const u64 nc = (u64)d.ncap + 1;
    // (obkeys must be cleared here, not in K5: k5's rows reference OBIP ranks only, but a K1a of the next
    // window may already be enqueued behind this kernel — same stream, so ordering is by launch order)
    if (!d.dh_g) for (u64 i = tid; i < nc * SG_DEG_REP; i += nt) { d.deg[i * SG_DEG_STRIDE] = 0; if (d.warm) d.deg2[i * SG_DEG_STRIDE] = 0; }    // (dh_g: no degree counters — k2_deg_hist rewrites every count it uses)
    for (u64 i = tid; i < nc; i += nt) d.cursor[i] = 0;
    for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_SUM_WORDS; i += nt) d.st_sum[i] = 0;
    for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_MAX_WORDS; i += nt) d.st_max[i] = 0;
    for (u64 i = tid; i <= d.obmask; i += nt) d.obkeys[i] = 0;
    // a window that is reset WITHOUT having been closed (sg_window_reset on an open window = discard): what the close
    // path would have consumed and re-armed — the per-workgroup K1 statistics, the overflow and alive lists
    for (u64 i = tid; i < (u64)SG_MAX_K1_WGS * WS_WORDS; i += nt) d.wgstat[i] = (i % WS_WORDS) == WS_TMIN ? ~0ull : 0ull;
    if (tid == 0) { d.ctr[C_OVF_N] = 0; d.ctr[C_ALIVE_N] = 0; d.ctr[C_ALIVE_DROP] = 0; }
    if (d.variant == 1 && d.ctr[C_EDGES_FOUND] > d.max_edges) {
        for (u64 i = tid; i <= d.emask; i += nt) {
            d.ekeys[i] = SG_EKEY_EMPTY;
            ulonglong2* a = reinterpret_cast<ulonglong2*>(d.eacc + (size_t)i * 4);
            a[0] = make_ulonglong2(0, 0); a[1] = make_ulonglong2(0, 0);
            if (d.hist) { uint4* hs = reinterpret_cast<uint4*>(d.hist_src + (size_t)i * SG_HIST_BINS); const uint4 z = make_uint4(0, 0, 0, 0); hs[0] = z; hs[1] = z; hs[2] = z; hs[3] = z; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4  sage_layer: h'_v = ReLU(b + h_v Ws + mean_{u in N_out(v)} h_u Wn)  on 16-node tiles.
//   gather-mean : one wave per node, lanes across features, 16 interleaved partial sums in the
//                 canonical order (neighbour i -> slot i % 16; slots combined 0..15; / deg).
//   dense       : 4 waves x v_mfma_f32_16x16x4_f32, k-ordered chain == the oracle's fmaf chain.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D[16 x 16] (+)= A[16 x K] * B[K x 16 cols jb..jb+15], C initialised with bias.  A in LDS (row
// stride lda), B = W[K][64] in global memory.  lane l: A[l&15][k=l>>4], B[k=l>>4][l&15];
// D reg r -> row (l>>4)*4 + r, col l&15.
template <int K>
__device__ __forceinline__ f32x4 dense_tile_mfma(const float* A, int lda, const float* __restrict__ W, int jb, f32x4 c) {
    const int l = threadIdx.x & 63, i = l & 15, kq = l >> 4;
#pragma unroll 4
    for (int kb = 0; kb < K / 4; kb++) {
        const float a = A[i * lda + kb * 4 + kq];
        const float b = W[(size_t)(kb * 4 + kq) * SG_F_HID + jb + i];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    return c;
}

// gather-mean of one node into dst[0..FI): executed by one wave.  Neighbour ids are fetched 64 at a
// time (one coalesced load) and broadcast by shuffle, so the 8 / 16 row loads of an unrolled step
// are independent and in flight together.  Summation order is the canonical one (slot = i % 16).
// Sum of the feature rows of neighbours [i_beg, i_end) of one node, one wave, written to dst[0..FI) (LDS).
// i_beg is a multiple of SG_MEAN_BLOCK and the range at most one block, so this is the block sum of the canonical
// mean: 16 interleaved slot sums (neighbour i -> slot i % 16, ascending i) combined in slot order.
// Lane layout: a lane loads four consecutive features (one 16-byte load) of one neighbour: c = lane % (FI/4) picks
// the features 4c..4c+3, g = lane / (FI/4) the neighbour inside a group of G = 256/FI; one load instruction fetches G
// whole rows and the 64/G loads of a 64-neighbour batch are all in flight together (one round trip per batch; the
// scalar-per-lane layout before needed two for FI = 32 and four for FI = 64).
// Neighbour i = G*a + g of a batch goes to slot i % 16 = G*(a % (16/G)) + g, i.e. accumulator a % (16/G) of group g.
// WIDE = loads in flight per 64-neighbour batch in the 16-byte layout (0: the one-feature-per-lane layout)
template <int FI, int WIDE>
__device__ __forceinline__ void gather_block_sum(const float* __restrict__ hin, const u32* __restrict__ nb, u32 i_beg, u32 i_end, float* dst) {
    if (WIDE == 0) {
        // one feature per lane (FI = 64), 16 row loads in flight: for the 1024-thread kernel, where the 16-byte
        // layout below needs more registers than there are (it spilled)
        const u32 lane = threadIdx.x & 63;
        float acc[16];
#pragma unroll
        for (int a = 0; a < 16; a++) acc[a] = 0.0f;
        u32 nxt = i_beg + lane < i_end ? nb[i_beg + lane] : 0u;
        for (u32 base = i_beg; base < i_end; base += 64) {
            const u32 cnt = i_end - base < 64 ? i_end - base : 64;
            const u32 my = nxt;
            nxt = base + 64 + lane < i_end ? nb[base + 64 + lane] : 0u;
            for (u32 i0 = 0; i0 < cnt; i0 += 16) {
#pragma unroll
                for (int a = 0; a < 16; a++) {
                    const u32 i = i0 + a;
                    const u32 id = __shfl(my, (int)i, 64);
                    if (i < cnt) acc[a] = acc[a] + hin[(size_t)id * 64 + lane];
                }
            }
        }
        float t = acc[0];
#pragma unroll
        for (int a = 1; a < 16; a++) t = t + acc[a];
        dst[lane] = t;
        return;
    }
    constexpr int C = FI / 4, G = 64 / C, NL = 64 / G, NA = 16 / G, NLC = NL < WIDE ? NL : WIDE;   // FI=32: 8 lanes per row, 8 groups, 8 loads, 2 accumulators
    const u32 lane = threadIdx.x & 63, c = lane % C, g = lane / C;
    const float4* __restrict__ h4 = reinterpret_cast<const float4*>(hin);
    float4 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) acc[a] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    u32 nxt = i_beg + lane < i_end ? nb[i_beg + lane] : 0u;             // ids of the next batch are fetched one batch ahead
    for (u32 base = i_beg; base < i_end; base += 64) {
        const u32 cnt = i_end - base < 64 ? i_end - base : 64;
        const u32 my = nxt;
        nxt = base + 64 + lane < i_end ? nb[base + 64 + lane] : 0u;
#pragma unroll
        for (int a0 = 0; a0 < NL; a0 += NLC) {                           // NLC loads in flight (register budget: 128 VGPRs at 1024 threads)
            float4 tmp[NLC];
#pragma unroll
            for (int a = 0; a < NLC; a++) {
                const u32 i = (u32)(G * (a0 + a)) + g;
                const u32 id = __shfl(my, (int)i, 64);
                tmp[a] = i < cnt ? h4[id * (u32)C + c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int a = 0; a < NLC; a++) if ((u32)(G * (a0 + a)) + g < cnt) {   // ascending neighbour index inside every slot
                float4& o = acc[(a0 + a) % NA];
                o.x = o.x + tmp[a].x; o.y = o.y + tmp[a].y; o.z = o.z + tmp[a].z; o.w = o.w + tmp[a].w;
            }
        }
    }
    // slots combined in slot order 0..15: slot s lives in group s % G, accumulator s / G (not unrolled: 64 shuffles
    // unrolled cost more registers than the kernel has)
    float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 1
    for (int sl = 0; sl < 16; sl++) {
        const int src = (int)c + C * (sl % G);
        float4 v = acc[0];
#pragma unroll
        for (int a = 1; a < NA; a++) if (sl / G == a) v = acc[a];
        const float x = __shfl(v.x, src, 64), y = __shfl(v.y, src, 64), z = __shfl(v.z, src, 64), w = __shfl(v.w, src, 64);
        if (sl == 0) t = make_float4(x, y, z, w);
        else { t.x = t.x + x; t.y = t.y + y; t.z = t.z + z; t.w = t.w + w; }
    }
    if (g == 0) { dst[4 * c] = t.x; dst[4 * c + 1] = t.y; dst[4 * c + 2] = t.z; dst[4 * c + 3] = t.w; }   // (dst is only 8-byte aligned in the tile)
}

// gather_block_sum for the stand-alone gather kernel: same sums in the same order, scheduled for the memory system.
//  * every row load is unconditional (index clamped to the block's last neighbour, the value dropped by a select), so
//    there is no branch per load: the batch's ids come out of NL back-to-back ds_bpermutes and its NL 16-byte row loads
//    go out back to back (the predicated version interleaved bpermute / wait / branch / load sixteen times);
//  * the 16 slot sums are combined through a wave-private LDS scratch (16 x FI floats): one store per accumulator,
//    then 16 independent 16-byte reads added in slot order by the lanes of group 0 — instead of 64 dependent shuffles.
template <int FI>
__device__ __forceinline__ void gather_block_sum2(const float* __restrict__ hin, const u32* __restrict__ nb, u32 i_beg, u32 i_end, float* dst, float* scr) {
    constexpr int C = FI / 4, G = 64 / C, NL = 64 / G, NA = 16 / G;
    const u32 lane = threadIdx.x & 63, c = lane % C, g = lane / C;
    const float4* __restrict__ h4 = reinterpret_cast<const float4*>(hin);
    float4 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) acc[a] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    u32 nxt = i_beg + lane < i_end ? nb[i_beg + lane] : 0u;             // ids of the next batch are fetched one batch ahead
    for (u32 base = i_beg; base < i_end; base += 64) {
        const u32 cnt = i_end - base < 64 ? i_end - base : 64;
        const u32 my = nxt;
        nxt = base + 64 + lane < i_end ? nb[base + 64 + lane] : 0u;
        u32 id[NL];
#pragma unroll
        for (int a = 0; a < NL; a++) { const u32 i = (u32)(G * a) + g; id[a] = __shfl(my, (int)(i < cnt ? i : cnt - 1), 64); }
        float4 tmp[NL];
#pragma unroll
        for (int a = 0; a < NL; a++) tmp[a] = h4[id[a] * (u32)C + c];
#pragma unroll
        for (int a = 0; a < NL; a++) {                                   // ascending neighbour index inside every slot
            const bool in = (u32)(G * a) + g < cnt;
            float4& o = acc[a % NA];
            o.x = in ? o.x + tmp[a].x : o.x; o.y = in ? o.y + tmp[a].y : o.y; o.z = in ? o.z + tmp[a].z : o.z; o.w = in ? o.w + tmp[a].w : o.w;
        }
    }
    // slot s = G*a + g  (accumulator a of group g)  ->  scr[s][4c..4c+3]
#pragma unroll
    for (int a = 0; a < NA; a++) reinterpret_cast<float4*>(scr + (size_t)(G * a + (int)g) * FI)[c] = acc[a];
    if (g == 0) {                                                        // (same wave: LDS operations of a wave are executed in order)
        float4 t = reinterpret_cast<const float4*>(scr)[c];
#pragma unroll
        for (int sl = 1; sl < 16; sl++) { const float4 v = reinterpret_cast<const float4*>(scr + (size_t)sl * FI)[c]; t.x = t.x + v.x; t.y = t.y + v.y; t.z = t.z + v.z; t.w = t.w + v.w; }
        dst[4 * c] = t.x; dst[4 * c + 1] = t.y; dst[4 * c + 2] = t.z; dst[4 * c + 3] = t.w;
    }
}
#define K4_HUB_BLOCKS 32         // block sums of a hub row kept in LDS per round
// Gather-mean as its own launch: 8 rows per 512-thread workgroup, a wave per row, two workgroups per CU — the gathers are
// L2-latency-bound and want waves in flight, the dense part wants 16-row tiles; fused in one kernel (round 1) a tile's waves
// waited for its longest row and a CU held one tile (C3: 57 + 99 us for the two layers).  Rows of more than one block: the
// blocks of the row are spread over the workgroup's 8 waves and added in block order, as before.  The result, mean[v][0..FI),
// is bit-identical to the fused version's (same gather_block_sum, same order of the block sums, one division).
#define K4G_ROWS 4              // rows per workgroup tile of k4_gather: one per wave.  (32 rows handed out by an LDS counter balanced the
                                // one-block rows better — 33 -> 28 us at C3 — but put several multi-block rows into one workgroup: 53 -> 62 us.
                                // Round 4, same box: 8 waves per workgroup 81-84 us for the two layers' K4, 4 waves 78.5; the 64-neighbour batch
                                // gathered in two / four parts — fewer registers, more waves — 83-87 / 94: the gather is bound by the bytes a
                                // wave keeps in flight, not by the waves.)
template <int FI>
__global__ __launch_bounds__(K4G_ROWS * 64) __attribute__((amdgpu_waves_per_eu(4))) void k4_gather(Dev d, const float* __restrict__ hin) {
    __shared__ __attribute__((aligned(16))) float scr_all[K4G_ROWS * 16 * FI];
    __shared__ __attribute__((aligned(16))) float part[K4G_ROWS * FI];
    const bool listed = d.world > 1 && d.ctr[C_ACT_L] != SG_ACT_NONE;
    const u32 N = listed ? (u32)d.ctr[C_ACT_L] : (u32)d.ctr[C_N_NODES];
    const u32 nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* scr = scr_all + wave * 16 * FI;
    // Rows of more than one block first: every 512-neighbour block of such a row is ONE work item of the list k2_rowptr built,
    // and the items are dealt to all waves of the launch (the 164 hub rows of C3 hold half of its edges: walked by the one
    // workgroup whose tile they fell into they were ~20 us of each launch).  The item's block sum goes to hub_part[item]; the
    // dense kernel adds a row's block sums in block order and divides — the same sums in the same order as before.
    {
        const u32 H = (u32)(d.ctr[C_HUB_ITEMS] < d.hub_cap ? d.ctr[C_HUB_ITEMS] : d.hub_cap);
        const u32 gw = blockIdx.x * K4G_ROWS + wave, nw = gridDim.x * K4G_ROWS;
        for (u32 it = gw; it < H; it += nw) {
            const uint2 x = d.hub_items[it];
            bool sk = false;
            if (d.world > 1) sk = owner_of_dense(d, x.x, nk, nl) != d.rank;      // (a hub row has out-edges: only its owner computes it)
            if (sk) continue;
            const u32 beg = d.rowptr[x.x], dg = d.rowptr[x.x + 1] - beg;
            const u32 i0 = x.y * SG_MEAN_BLOCK, i1 = i0 + SG_MEAN_BLOCK < dg ? i0 + SG_MEAN_BLOCK : dg;
            gather_block_sum2<FI>(hin, d.col + beg, i0, i1, d.hub_part + (size_t)it * SG_F_HID, scr);
        }
    }
    for (u32 tile = blockIdx.x; tile * K4G_ROWS < N; tile += gridDim.x) {
        const u32 i = tile * K4G_ROWS + wave;
        bool sk = i >= N;
        const u32 v = sk ? 0u : (listed ? d.act_l[i] : i);
        if (!sk && !listed && d.world > 1) {
            const bool has_out = d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + ST_OUT_DEG] != 0;
            sk = has_out && owner_of_dense(d, v, nk, nl) != d.rank;
        }
        if (!sk) {
            const u32 beg = d.rowptr[v];
            const u32 deg = d.rowptr[v + 1] - beg;
            float* dst = part + wave * FI;
            if (deg && deg <= SG_MEAN_BLOCK) gather_block_sum2<FI>(hin, d.col + beg, 0, deg, dst, scr);
            if (deg <= SG_MEAN_BLOCK && lane < FI) d.nmean[(size_t)v * SG_F_HID + lane] = deg ? dst[lane] / (float)deg : 0.0f;   // (same wave wrote dst)
        }
    }
}

// 16-node tiles, 1024 threads: in the gather phase every wave owns one node of the tile (the rows
// follow a power law, so per-node parallelism is what bounds this kernel); the dense phase runs on
// the first 4 waves.  With PROJ the tile's fresh h rows are immediately projected to the score
// head's P = b1 + h Wu and Q = h Wv (last layer, unsharded), saving a launch.
// NT = 1024 (one wave per tile row) for the 32-feature first layer; NT = 512 (a wave takes two rows) for the
// 64-feature hidden layers: twice the registers per lane, so all 16 loads of a batch in the 16-byte layout are in
// flight (a quarter of the round trips on hub rows; C3 layer 2: 369 us before).
// PRE: the neighbour means were computed by k4_gather (d.nmean): phase 1 only copies rows, the hub loop is gone.
template <int FI, bool USE_MFMA, bool PROJ, int NT, bool PRE = false>
__global__ __launch_bounds__(NT) void k4_sage_layer(Dev d, const float* __restrict__ hin, float* __restrict__ hout, const float* __restrict__ Wl, const float* __restrict__ Wh) {
    constexpr int LDA = 2 * FI + 2;                               // +2 floats: conflict-free A-fragment reads
    constexpr int LDH = SG_F_HID + 2;
    __shared__ float A[16 * LDA];
    __shared__ float H[PROJ ? 16 * LDH : 1];
    __shared__ u32 skip[16];
    __shared__ u32 vid[16], tdeg[16];
    __shared__ float hub[PRE ? 1 : K4_HUB_BLOCKS * FI];
    // world > 1: walk the shard's active list (local sources + local leaf destinations; the rows of remote
    // sources arrive by halo exchange); unsharded: every node
    const bool listed = !PROJ && d.world > 1 && d.ctr[C_ACT_L] != SG_ACT_NONE;   // lists are built with the halo requests
    const u32 N = listed ? (u32)d.ctr[C_ACT_L] : (u32)d.ctr[C_N_NODES];
    const u32 nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    constexpr u32 NW = NT / 64;
    constexpr int WIDE = FI == 32 ? 8 : (NT <= 512 ? 16 : 0);
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* __restrict__ bias = Wl + 2 * FI * SG_F_HID;
    for (u32 tile = blockIdx.x; tile * 16 < N; tile += gridDim.x) {
        const u32 v0 = tile * 16;
        for (u32 r = wave; r < 16; r += NW) {   // phase 1: self row + gather-mean, one wave per tile row
            bool sk = v0 + r >= N;
            const u32 v = sk ? 0u : (listed ? d.act_l[v0 + r] : v0 + r);
            if (!sk && !listed && d.world > 1) {                     // no list this window: walk all nodes, skip what an owner computes
                const bool has_out = d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + ST_OUT_DEG] != 0;
                sk = has_out && owner_of_dense(d, v, nk, nl) != d.rank;
            }
            float* row = A + r * LDA;
            u32 deg = 0;
            if (sk) { for (u32 k = lane; k < 2 * FI; k += 64) row[k] = 0.0f; }
            else {
                for (u32 k = lane; k < FI; k += 64) row[k] = hin[(size_t)v * FI + k];
                const u32 beg = d.rowptr[v];
                deg = d.rowptr[v + 1] - beg;
                if constexpr (PRE) {
                    if (deg > SG_MEAN_BLOCK) {                       // a hub row: its block sums (k4_gather's work items) in block order, one division
                        const u32 nblk = (deg + SG_MEAN_BLOCK - 1) / SG_MEAN_BLOCK, ib = d.hub_base[v];
                        for (u32 k = lane; k < FI; k += 64) {
                            float total = d.hub_part[(size_t)ib * SG_F_HID + k];
                            for (u32 j = 1; j < nblk; j++) total = total + d.hub_part[(size_t)(ib + j) * SG_F_HID + k];
                            row[FI + k] = total / (float)deg;
                        }
                    } else for (u32 k = lane; k < FI; k += 64) row[FI + k] = deg ? d.nmean[(size_t)v * SG_F_HID + k] : 0.0f;
                    deg = 0;                                         // (nothing left for the hub loop)
                } else {
                // block 0 here (one wave per row, all rows at once); the further blocks of a hub row below
                if (deg) gather_block_sum<FI, WIDE>(hin, d.col + beg, 0, deg < SG_MEAN_BLOCK ? deg : SG_MEAN_BLOCK, row + FI);
                if (lane < FI) {                                     // (same wave wrote row[FI..): ordered by the LDS counter)
                    const float t = deg ? row[FI + lane] : 0.0f;
                    row[FI + lane] = deg > SG_MEAN_BLOCK ? t : (deg ? t / (float)deg : 0.0f);
                }
                }
            }
            if (lane == 0) { skip[r] = sk ? 1u : 0u; vid[r] = v; tdeg[r] = deg; }
        }
        __syncthreads();
        // hub rows (more than one block): the blocks of a row are spread over the 16 waves, the block sums are
        // then added in block order by one wave — a 3000-neighbour row no longer serialises on a single wave
        if constexpr (!PRE) for (u32 r = 0; r < 16; r++) {
            const u32 deg = tdeg[r];
            if (deg <= SG_MEAN_BLOCK) continue;                      // uniform
            const u32 v = vid[r], beg = d.rowptr[v], nblk = (deg + SG_MEAN_BLOCK - 1) / SG_MEAN_BLOCK;
            float total = (wave == 0 && lane < FI) ? A[r * LDA + FI + lane] : 0.0f;   // block 0, from above (wave 0, lanes < FI)
            for (u32 b0 = 1; b0 < nblk; b0 += K4_HUB_BLOCKS) {
                const u32 bn = nblk - b0 < K4_HUB_BLOCKS ? nblk - b0 : K4_HUB_BLOCKS;
                for (u32 j = wave; j < bn; j += NW) {
                    const u32 i0 = (b0 + j) * SG_MEAN_BLOCK, i1 = i0 + SG_MEAN_BLOCK < deg ? i0 + SG_MEAN_BLOCK : deg;
                    gather_block_sum<FI, WIDE>(hin, d.col + beg, i0, i1, hub + j * FI);
                }
                __syncthreads();
                if (wave == 0 && lane < FI) for (u32 j = 0; j < bn; j++) total = total + hub[j * FI + lane];
                __syncthreads();
            }
            if (wave == 0 && lane < FI) A[r * LDA + FI + lane] = total / (float)deg;
        }
        __syncthreads();
        // phase 2: dense 16 x 64 on waves 0..3, wave w -> columns 16w..16w+15
        if (wave < 4) {
            if (USE_MFMA) {
                const int jb = wave * 16, i = lane & 15;
                const float bj = bias[jb + i];
                f32x4 c = { bj, bj, bj, bj };
                c = dense_tile_mfma<2 * FI>(A, LDA, Wl, jb, c);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const u32 row = (lane >> 4) * 4 + r;
                    const float hv = c[r] > 0.0f ? c[r] : 0.0f;
                    if (!skip[row]) hout[(size_t)vid[row] * SG_F_HID + jb + i] = hv;
                    if (PROJ) H[row * LDH + jb + i] = hv;
                }
            } else {
                const u32 row = (threadIdx.x & 255) >> 4, jq = (threadIdx.x & 15) * 4;
                float acc[4];
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] = bias[jq + c];
                for (int k = 0; k < 2 * FI; k++) {
                    const float a = A[row * LDA + k];
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[c] = fmaf(a, Wl[(size_t)k * SG_F_HID + jq + c], acc[c]);
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float hv = acc[c] > 0.0f ? acc[c] : 0.0f;
                    if (!skip[row]) hout[(size_t)vid[row] * SG_F_HID + jq + c] = hv;
                    if (PROJ) H[row * LDH + jq + c] = hv;
                }
            }
        }
        if (PROJ) {
            __syncthreads();
            // waves 0..3 -> P columns, waves 4..7 -> Q columns
            const float* __restrict__ Wu = Wh; const float* __restrict__ Wv = Wh + SG_F_HID * SG_F_HID;
            const float* __restrict__ b1 = Wv + SG_F_HID * SG_F_HID + SG_F_EDGE * SG_F_HID;
            if (wave < 8) {
                const bool isq = wave >= 4;
                const int jb = (wave & 3) * 16, i = lane & 15;
                float* dst = isq ? d.Q : d.P;
                const float* __restrict__ Wm = isq ? Wv : Wu;
                if (USE_MFMA) {
                    const float bj = isq ? 0.0f : b1[jb + i];
                    f32x4 c = { bj, bj, bj, bj };
                    c = dense_tile_mfma<SG_F_HID>(H, LDH, Wm, jb, c);
#pragma unroll
                    for (int r = 0; r < 4; r++) { const u32 row = (lane >> 4) * 4 + r; if (v0 + row < N) dst[(size_t)(v0 + row) * SG_F_HID + SG_PQ_POS(jb + i)] = c[r]; }
                } else {
                    // VALU twin: lane -> (row = lane >> 2, 4 columns)
                    const u32 row = lane >> 2, jq = jb + (lane & 3) * 4;
                    float acc[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[c] = isq ? 0.0f : b1[jq + c];
                    for (int k = 0; k < (int)SG_F_HID; k++) {
                        const float a = H[row * LDH + k];
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[c] = fmaf(a, Wm[(size_t)k * SG_F_HID + jq + c], acc[c]);
                    }
                    if (v0 + row < N)
#pragma unroll
                        for (int c = 0; c < 4; c++) dst[(size_t)(v0 + row) * SG_F_HID + SG_PQ_POS(jq + c)] = acc[c];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K5  edge_score: P = b1 + h Wu, Q = h Wv per node (MFMA), then per edge
//     s = sigmoid(b2 + tree_sum_j( ReLU(P[u][j] + Q[v][j] + sum_k e[k] We[k][j]) * w2[j] )).
// ------------------------------------------------------------------------------------------------
template <bool USE_MFMA>
__global__ __launch_bounds__(256) void k5_node_proj(Dev d, const float* __restrict__ hL, const float* __restrict__ Wh) {
    constexpr int LDA = SG_F_HID + 2;
    __shared__ float A[16 * LDA];
    __shared__ u32 vid[16];
    const bool listed = d.world > 1 && d.ctr[C_ACT_L] != SG_ACT_NONE;   // only the endpoints of this shard's edges
    const u32 N = listed ? (u32)d.ctr[C_ACT_P] : (u32)d.ctr[C_N_NODES];
    const float* __restrict__ Wu = Wh; const float* __restrict__ Wv = Wh + SG_F_HID * SG_F_HID;
    const float* __restrict__ b1 = Wv + SG_F_HID * SG_F_HID + SG_F_EDGE * SG_F_HID;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (u32 tile = blockIdx.x; tile * 16 < N; tile += gridDim.x) {
        const u32 v0 = tile * 16;
        if (threadIdx.x < 16) vid[threadIdx.x] = v0 + threadIdx.x < N ? (listed ? d.act_p[v0 + threadIdx.x] : v0 + threadIdx.x) : 0u;
        __syncthreads();
        for (u32 idx = threadIdx.x; idx < 16 * SG_F_HID; idx += 256) {
            const u32 r = idx >> 6, k = idx & 63;
            A[r * LDA + k] = (v0 + r < N) ? hL[(size_t)vid[r] * SG_F_HID + k] : 0.0f;
        }
        __syncthreads();
        if (USE_MFMA) {
            const int jb = wave * 16, i = lane & 15;
            const float bj = b1[jb + i];
            f32x4 p = { bj, bj, bj, bj }, q = { 0.0f, 0.0f, 0.0f, 0.0f };
            p = dense_tile_mfma<SG_F_HID>(A, LDA, Wu, jb, p);
            q = dense_tile_mfma<SG_F_HID>(A, LDA, Wv, jb, q);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u32 row = (lane >> 4) * 4 + r;
                if (v0 + row < N) { d.P[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jb + i)] = p[r]; d.Q[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jb + i)] = q[r]; }
            }
        } else {
            const u32 row = threadIdx.x >> 4, jq = (threadIdx.x & 15) * 4;
            float p[4], q[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { p[c] = b1[jq + c]; q[c] = 0.0f; }
            for (int k = 0; k < (int)SG_F_HID; k++) {
                const float a = A[row * LDA + k];
#pragma unroll
                for (int c = 0; c < 4; c++) { p[c] = fmaf(a, Wu[(size_t)k * SG_F_HID + jq + c], p[c]); q[c] = fmaf(a, Wv[(size_t)k * SG_F_HID + jq + c], q[c]); }
            }
            if (v0 + row < N)
#pragma unroll
                for (int c = 0; c < 4; c++) { d.P[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jq + c)] = p[c]; d.Q[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jq + c)] = q[c]; }
        }
        __syncthreads();
    }
}

// One wave scores 4 edges per step: 16 lanes per edge, lane q of a group owns hidden units q + 16 m (m = 0..3).
//   t_j = P[u][j] + Q[v][j] + sum_k e_k We[k][j] (fmaf chain over k), ReLU, * w2[j]
//   sum over j in the canonical butterfly order (strides 32, 16, 8, 4, 2, 1): strides 32 and 16 pair units of
//   the SAME lane (j ^ 32 <-> m ^ 2, j ^ 16 <-> m ^ 1), strides 8..1 are DPP steps inside the group's row of 16 lanes — the
//   same additions in the same order as one lane per unit (fp32 addition commutes bitwise), at a quarter of the
//   instructions per edge.  Lane 0 of a group writes the edge's row.
static_assert(sizeof(sg_edge_out) == 64 && offsetof(sg_edge_out, sum_ns) == 0 && offsetof(sg_edge_out, max_ns) == 8 && offsetof(sg_edge_out, sumsq_us) == 16 &&
              offsetof(sg_edge_out, from_ref) == 24 && offsetof(sg_edge_out, to_ref) == 28 && offsetof(sg_edge_out, count) == 32 && offsetof(sg_edge_out, err_count) == 36 &&
              offsetof(sg_edge_out, score) == 40 && offsetof(sg_edge_out, lat_z) == 44 && offsetof(sg_edge_out, err_ratio) == 48 && offsetof(sg_edge_out, alive) == 52 &&
              offsetof(sg_edge_out, p50_us) == 56 && offsetof(sg_edge_out, p99_us) == 60, "k5_edge_score writes a row as eight 8-byte words");
template <bool RESET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k5_edge_score(Dev d, const float* __restrict__ Wh) {
    const u32 E = (u32)d.ctr[C_N_EDGES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const float* __restrict__ We = Wh + 2 * SG_F_HID * SG_F_HID;
    const float* __restrict__ w2 = We + SG_F_EDGE * SG_F_HID + SG_F_HID;
    const float b2 = w2[SG_F_HID];
    const u32 lane = threadIdx.x & 63, q = lane & 15, g = lane >> 4;
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    float we[SG_F_EDGE][4], w2r[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        w2r[m] = w2[q + 16 * m];
#pragma unroll
        for (int k = 0; k < (int)SG_F_EDGE; k++) we[k][m] = We[k * SG_F_HID + q + 16 * m];
    }
    // Two steps of four edges per iteration, and the endpoints of the NEXT iteration's edges are fetched while this one's
    // rows are gathered: the dependent chain per iteration is one round trip (the gathers), not two (ids, then gathers).
    // What the 16 lanes of a group need in common is loaded ONCE per group and spread with DPP row_newbcast (a v_mov per value):
    // the two edges' feature vectors are one dword per lane (lanes 0..7 edge A's e_0..e_7, lanes 8..15 edge B's) instead of four
    // 16-byte loads that return the same 32 bytes to all sixteen lanes, the four endpoint ids one dword in lanes 0..3 instead of
    // four loads — 9 memory instructions and 5.5 KiB returned per wave and iteration instead of 15 and 10 KiB (the kernel is
    // bound by the vector-memory pipe, not by HBM: P and Q are L2-resident).
    auto step = [&](const float4 P4, const float4 Q4, const float (&ek)[SG_F_EDGE]) -> float {
        const float pq[4] = {P4.x + Q4.x, P4.y + Q4.y, P4.z + Q4.z, P4.w + Q4.w};
        float r[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            float x = pq[m];
#pragma unroll
            for (int k = 0; k < (int)SG_F_EDGE; k++) x = fmaf(ek[k], we[k][m], x);
            x = x > 0.0f ? x : 0.0f;
            r[m] = x * w2r[m];
        }
        float sum = (r[0] + r[2]) + (r[1] + r[3]);                  // strides 32, then 16
        sum = sum + xor_partner_f32(sum, 8); sum = sum + xor_partner_f32(sum, 4);
        sum = sum + xor_partner_f32(sum, 2); sum = sum + xor_partner_f32(sum, 1);
        return sum;                                                  // (every lane of the 16 holds it)
    };
    static_assert(SG_F_EDGE == 8, "k5_edge_score spreads two 8-float edge feature vectors over a DPP row of 16 lanes");
    const u32* __restrict__ idsrc = (q & 1u) ? d.col : d.csr_from;   // lane q & 3 of a group: from(A), to(A), from(B), to(B)
    // The ROWS of an iteration's eight edges are written by the whole wave: 8 x 64 bytes = 64 lanes x 8 bytes, lane l holds
    // 8-byte word l % 8 of edge l / 8 — one fully coalesced store per iteration instead of four 16-byte stores from one lane in
    // sixteen per step (whose ~100 instructions of row assembly ran with 4 of 64 lanes active).
    //   word 0..2 sum_ns, max_ns, sumsq_us = accumulators 1..3; word 3 from_ref | to_ref; word 4 count | err = accumulator 0;
    //   word 5 score | lat_z; word 6 err_ratio | alive; word 7 p50_us | p99_us
    const u32 wk = lane & 7u, we8 = lane >> 3;
    const u32* __restrict__ srcA = wk == 3 ? d.csr_from : reinterpret_cast<const u32*>(d.errr);
    const u32* __restrict__ srcB = wk == 3 ? d.col : (wk == 5 ? reinterpret_cast<const u32*>(d.latz) : d.alive_csr);
    const u32 accj = wk < 3 ? wk + 1 : 0u;
    const int srcl = (int)(((we8 & 3u) << 4) | (we8 < 4 ? 0u : 8u));     // lane that holds the score sum of this lane's edge: group (edge & 3), its
                                                                         // lower half for the first step's four edges, its upper half for the second's
    if (E) {
        const u32 stride = nw * 8, last = E - 1;
        u32 pa = wave * 8 + g, pb = pa + 4;                          // this iteration's two edges of the lane group (clamped when beyond E)
        // Round 4: TWO iterations in flight.  A wave runs ~30 iterations at C3 and an iteration was one exposed round trip (the gathers:
        // ~2 us) beside ~0.4 us of arithmetic — 73 us of which 60 were latency at four waves per SIMD.  Now the gathers, the feature
        // dword and the row words of iteration i + 1 are issued BEFORE iteration i is computed (a second register set: 21 dwords), its
        // endpoint ids having been fetched an iteration earlier still; loads return in order, so waiting for set i does not wait for
        // set i + 1.  (The set beyond the last iteration is loaded from clamped addresses and dropped.)
        struct K5Set { float4 PA, QA, PB, QB; u32 ew, wa, wb; u64 wacc; };
        auto ids_of = [&](u32 xa, u32 xb) -> u32 { const u32 px = (q & 2u) ? xb : xa; return idsrc[px < E ? px : last]; };
        auto issue = [&](u32 idw_, u32 xa, u32 xb, u32 x0, K5Set& S) {
            u32 ua = dpp32b<0x150>(idw_), va = dpp32b<0x151>(idw_), ub = dpp32b<0x152>(idw_), vb = dpp32b<0x153>(idw_);
            if SG_ABL(d, 0x1000u) { ua &= 15u; va &= 15u; ub &= 15u; vb &= 15u; }   // (diagnostic: gathers that hit the L1)
            const u32 ca = xa < E ? xa : last, cb = xb < E ? xb : last;
            S.PA = reinterpret_cast<const float4*>(d.P + (size_t)ua * SG_F_HID)[q]; S.QA = reinterpret_cast<const float4*>(d.Q + (size_t)va * SG_F_HID)[q];
            S.PB = reinterpret_cast<const float4*>(d.P + (size_t)ub * SG_F_HID)[q]; S.QB = reinterpret_cast<const float4*>(d.Q + (size_t)vb * SG_F_HID)[q];
            S.ew = __float_as_uint(d.efeat[(size_t)(q < 8 ? ca : cb) * SG_F_EDGE + (q & 7u)]);
            // what this lane's row word is made of: fetched beside the gathers
            const u32 er_ = x0 + we8, ec_ = er_ < E ? er_ : last;
            S.wacc = d.acc_csr[(size_t)ec_ * 4 + accj];
            S.wa = srcA[ec_]; S.wb = srcB[ec_];
        };
        // One iteration: `cur` is computed and stored, `nxt` issued; the two sets ALTERNATE between the two calls of the loop body — rotating
        // them through copies at the back edge made the compiler wait for the set in flight there (a v_mov needs its source loaded).
        auto iter = [&](K5Set& cur, K5Set& nxt, const u32 idw_next, u32& idw_after, const u32 p0) {
            const u32 na = pa + stride, nb = pb + stride;
            // the endpoints of the iteration after the next go out FIRST: they are then older than the gathers issued below, and the next
            // iteration's wait for them does not wait for those gathers (issued the other way round, it was a vmcnt(0) at the loop top)
            idw_after = ids_of(na + stride, nb + stride);
            issue(idw_next, na, nb, p0 + stride, nxt);
            __builtin_amdgcn_sched_barrier(0);                       // (the scheduler moved the arithmetic of `cur` above these loads)
            const float4 PA = cur.PA, QA = cur.QA, PB = cur.PB, QB = cur.QB;
            const u32 ew = cur.ew, wa = cur.wa, wb = cur.wb; const u64 wacc = cur.wacc;
            const u32 er = p0 + we8, ec = er < E ? er : last;
            const float eka[SG_F_EDGE] = {__uint_as_float(dpp32b<0x150>(ew)), __uint_as_float(dpp32b<0x151>(ew)), __uint_as_float(dpp32b<0x152>(ew)), __uint_as_float(dpp32b<0x153>(ew)),
                                          __uint_as_float(dpp32b<0x154>(ew)), __uint_as_float(dpp32b<0x155>(ew)), __uint_as_float(dpp32b<0x156>(ew)), __uint_as_float(dpp32b<0x157>(ew))};
            const float ekb[SG_F_EDGE] = {__uint_as_float(dpp32b<0x158>(ew)), __uint_as_float(dpp32b<0x159>(ew)), __uint_as_float(dpp32b<0x15A>(ew)), __uint_as_float(dpp32b<0x15B>(ew)),
                                          __uint_as_float(dpp32b<0x15C>(ew)), __uint_as_float(dpp32b<0x15D>(ew)), __uint_as_float(dpp32b<0x15E>(ew)), __uint_as_float(dpp32b<0x15F>(ew))};
            const float sa = step(PA, QA, eka);
            const float sb = step(PB, QB, ekb);
            const float mysum = __shfl(q < 8 ? sa : sb, srcl, 64);     // ONE shuffle executed by all lanes (two under a select were sunk into exec-masked
                                                                       // branches by the compiler: ds_bpermute returns 0 for an inactive source lane)
            u64 val = wacc;                                          // words 0..2 and 4
            if (wk == 3) val = (u64)ref_of_dense(wa, nk, nl) | ((u64)ref_of_dense(wb, nk, nl) << 32);
            else if (wk == 5) { const float logit = mysum + b2; val = (u64)__float_as_uint(1.0f / (1.0f + expf(-logit))) | ((u64)wb << 32); }
            else if (wk == 6) val = (u64)wa | ((u64)wb << 32);
            else if (wk == 7) {
                val = 0;
                if (d.hist) {                                        // percentiles off the log2 histogram (include/servicegraph.h)
                    const ulonglong2* __restrict__ ac = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)ec * 4);
                    const u32 count = (u32)(ac[0].x & 0xFFFFFFFFull); const u64 max_ns = ac[1].x;
                    if (count) {
                        const uint4* hp = reinterpret_cast<const uint4*>(d.hist_csr + (size_t)ec * SG_HIST_BINS);
                        const uint4 h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3];
                        const u32 hb[SG_HIST_BINS] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
                        u64 r50 = ((u64)count * 50 + 99) / 100, r99 = ((u64)count * 99 + 99) / 100;
                        r50 = r50 ? r50 : 1; r99 = r99 ? r99 : 1;
                        u64 cum = 0; u32 b50 = SG_HIST_BINS - 1, b99 = SG_HIST_BINS - 1; bool f50 = false, f99 = false;
#pragma unroll
                        for (u32 b = 0; b < SG_HIST_BINS; b++) { cum += hb[b]; if (!f50 && cum >= r50) { b50 = b; f50 = true; } if (!f99 && cum >= r99) { b99 = b; f99 = true; } }
                        u64 e50 = b50 == SG_HIST_BINS - 1 ? max_ns : (1ull << (17 + b50)), e99 = b99 == SG_HIST_BINS - 1 ? max_ns : (1ull << (17 + b99));
                        e50 = e50 > max_ns ? max_ns : e50; e99 = e99 > max_ns ? max_ns : e99;
                        e50 /= 1000ull; e99 /= 1000ull;
                        val = (u64)(e50 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)e50) | ((u64)(e99 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)e99) << 32);
                    }
                }
            }
            // The store as a BUFFER store on the wave's 512 bytes of this iteration (p0 is wave-uniform): rows beyond E are dropped by
            // the resource's range check, not by a branch — a branch around the store is a join for the compiler's vmcnt bookkeeping,
            // and the next half-iteration's wait for its endpoint ids then also waited for the first gather of the set in flight.
            {
                const u32 p0u = (u32)__builtin_amdgcn_readfirstlane((int)p0);
                const u32 nrow = SG_ABL(d, 0x2000u) ? 0u : (E - p0u < 8u ? E - p0u : 8u);
                const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u64*>(d.rows) + (size_t)p0u * 8, 0, (int)(nrow * 64u), 0x00020000);
                v2u_t dv; dv.x = (u32)val; dv.y = (u32)(val >> 32);
                __builtin_amdgcn_raw_buffer_store_b64(dv, rr_, lane * 8u, 0, 0);
            }
            pa = na; pb = nb;
        };
        K5Set A, B;
        u32 i1 = ids_of(pa + stride, pb + stride), i2;               // the endpoints of iteration 1 ...
        { const u32 id0 = ids_of(pa, pb); issue(id0, pa, pb, wave * 8, A); }   // ... in flight before iteration 0's gathers
        for (u32 p0 = wave * 8; p0 < E; p0 += 2 * stride) {
            iter(A, B, i1, i2, p0);
            if (p0 + stride >= E) break;                             // (uniform)
            iter(B, A, i2, i1, p0 + stride);
        }
    }
    if (RESET) {
        // Window reset folded into the last kernel of the pipeline (nothing after it reads these arrays;
        // the counters stay: sg_window_read / the next kc_prepare consume them).
        const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, nt = (u64)gridDim.x * 256;
        const u64 nc = (u64)d.ncap + 1;
        if (!d.dh_g) for (u64 i = tid; i < nc * SG_DEG_REP; i += nt) { d.deg[i * SG_DEG_STRIDE] = 0; if (d.warm) d.deg2[i * SG_DEG_STRIDE] = 0; }
        for (u64 i = tid; i < nc; i += nt) d.cursor[i] = 0;
        for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_SUM_WORDS; i += nt) d.st_sum[i] = 0;
        for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_MAX_WORDS; i += nt) d.st_max[i] = 0;
        for (u64 i = tid; i <= d.obmask; i += nt) d.obkeys[i] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// K6  halo: which remote rows this shard needs, and pack / unpack of feature rows.
// ------------------------------------------------------------------------------------------------
// thread per node: v is in the halo if it is the destination of a local edge (local in-degree > 0
// is tracked in `cursor`, reused as a mark array), is not owned here, and has out-edges somewhere.
__global__ __launch_bounds__(256) void k6_halo_mark(Dev d) {
    const u32 E = (u32)d.ctr[C_N_EDGES];
    for (u32 p = blockIdx.x * 256 + threadIdx.x; p < E; p += gridDim.x * 256) d.cursor[d.col[p]] = 0xFFFFFFFFu;
}
__global__ __launch_bounds__(256) void k6_halo_build(Dev d, u32* ids, u32 cap, u32* counts) {
    // single workgroup; output grouped by owner shard, ascending dense id inside a group, so every
    // run (and every shard, for the ids it is asked for) sees the same lists.
    const u32 N = (u32)d.ctr[C_N_NODES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    __shared__ u32 part[8][256];
    __shared__ u32 base[8];
    const u32 W = d.world < 8 ? d.world : 8;
    const u32 per = (N + 255) / 256, beg = threadIdx.x * per, end = beg + per < N ? beg + per : N;
    u32 c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 v = beg; v < end; v++) {
        if (d.cursor[v] != 0xFFFFFFFFu || d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + ST_OUT_DEG] == 0) continue;
        const u32 o = owner_of_dense(d, v, nk, nl);
        if (o == d.rank) continue;
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] += (o == (u32)k);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) part[k][threadIdx.x] = c[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (u32 k = 0; k < W; k++) {
            base[k] = run;
            u32 tot = 0;
            for (int t = 0; t < 256; t++) { const u32 x = part[k][t]; part[k][t] = run + tot; tot += x; }
            counts[k] = (run + tot <= cap) ? tot : (run < cap ? cap - run : 0);
            run += tot;
        }
    }
    __syncthreads();
    u32 pos[8];
#pragma unroll
    for (int k = 0; k < 8; k++) pos[k] = part[k][threadIdx.x];
    for (u32 v = beg; v < end; v++) {
        if (d.cursor[v] != 0xFFFFFFFFu || d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + ST_OUT_DEG] == 0) continue;
        const u32 o = owner_of_dense(d, v, nk, nl);
        if (o == d.rank) continue;
#pragma unroll
        for (int k = 0; k < 8; k++) if (o == (u32)k) { if (pos[k] < cap) ids[pos[k]] = v; pos[k]++; }
    }
}
// rows[i][:] = feat[ids[i]][:]   (16 lanes x float4 per 64-float row)
__global__ __launch_bounds__(256) void k6_pack(const float* __restrict__ feat, const u32* __restrict__ ids, u32 n, float* __restrict__ rows) {
    for (u32 t = blockIdx.x * 256 + threadIdx.x; t < n * 16; t += gridDim.x * 256) {
        const u32 i = t >> 4, q = t & 15;
        reinterpret_cast<float4*>(rows)[(size_t)i * 16 + q] = reinterpret_cast<const float4*>(feat)[(size_t)ids[i] * 16 + q];
    }
}
__global__ __launch_bounds__(256) void k6_unpack(float* __restrict__ feat, const u32* __restrict__ ids, u32 n, const float* __restrict__ rows) {
    for (u32 t = blockIdx.x * 256 + threadIdx.x; t < n * 16; t += gridDim.x * 256) {
        const u32 i = t >> 4, q = t & 15;
        reinterpret_cast<float4*>(feat)[(size_t)ids[i] * 16 + q] = reinterpret_cast<const float4*>(rows)[(size_t)i * 16 + q];
    }
}

// Per-node flags of the halo / active-list sweep: bit 0 = destination of a local edge, bit 1 = source of one,
// bit 2 = has out-edges somewhere (global out-degree, after the statistics all-reduce), bits 3.. = owner + 1
// when the node is a halo node (remote owner, out-edges, local destination), else 0.
__device__ __forceinline__ u32 node_flags(const Dev& d, u32 v, u32 nk, u32 nl) {
    const bool dst = d.cursor[v] == 0xFFFFFFFFu, src = d.rowptr[v + 1] != d.rowptr[v];
    const bool has_out = d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + ST_OUT_DEG] != 0;
    u32 f = (dst ? 1u : 0u) | (src ? 2u : 0u) | (has_out ? 4u : 0u);
    if (dst && has_out) { const u32 o = owner_of_dense(d, v, nk, nl); if (o != d.rank) f |= (o + 1) << 3; }
    return f;
}
#define K6_FLAGS_LDS 49152       // nodes whose flags fit the LDS staging of the list builder

// The halo request lists and the shard's active node lists (world > 1) in one sweep over the nodes by a
// 1024-thread workgroup.  With N nodes in the map and only ~N/world of them touched here, the layer and
// projection kernels must not walk all N (that would undo weak scaling):
//   act_l: nodes whose layer output is computed here = local sources + local destinations without out-edges anywhere
//   act_p: nodes whose score projections are needed here = the endpoints of the local edges
//   req[k]: halo nodes owned by shard k (k < 8), ascending — every shard builds the same lists
// Flags are first staged in LDS with coalesced loads (thread t, nodes t, t + 1024, ...); the ordered passes
// then give thread t the contiguous chunk [beg, end) so that thread order is ascending node order.
// The staged form (N <= K6_FLAGS_LDS, every map so far): ordered compaction by WAVES, not by threads.  Wave w owns the contiguous node
// block [w * per_w, (w + 1) * per_w) and walks it 64 nodes a step; a list's position of node v = the wave's base (one exchange of
// the sixteen waves' totals through LDS) + the members in the wave's earlier steps + the members among the lower lanes of this
// step (ballot + popcount) — ascending by construction, the 64 lanes of a step write adjacent entries, and the only barriers are
// the one behind the flag staging and the pair around the totals.  (A thread per contiguous 15-node chunk — the form below, kept
// for maps beyond the LDS staging — was fifteen serial rounds of scattered 4-byte stores per list: 53 us of a C4 shard's window.)
template <bool REQ>
__device__ __forceinline__ void build_lists_staged(const Dev& d, u32* req, u32 capp, unsigned char* fl, u32* wsum) {
    [[maybe_unused]] constexpr bool want_req = REQ;
    const u32 N = (u32)d.ctr[C_N_NODES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const u32 W = d.world < 8 ? d.world : 8;
    // the flags of eight nodes per thread and trip, every load of the trip issued before any of them is used: node_flags() has a
    // branch (the owner is only computed for halo candidates) behind which the compiler parks the next node's loads — one node
    // after the other was two dependent round trips x 15 nodes per thread, most of this kernel's 45-50 us
    const u32 nkl = nk + nl, nobs = N > nkl ? N - nkl : 0u;
    for (u32 v0 = threadIdx.x; v0 < N; v0 += 8192) {
        u32 cur[8], r0[8], r1[8], obi[8]; u64 od[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const u32 v = v0 + q * 1024 < N ? v0 + q * 1024 : N - 1;
            cur[q] = d.cursor[v]; r0[q] = d.rowptr[v]; r1[q] = d.rowptr[v + 1];
            od[q] = d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + ST_OUT_DEG];
            obi[q] = nobs ? d.ob_sorted[v >= nkl ? v - nkl : 0u] : 0u;         // (only used for an outbound-ip node)
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const u32 v = v0 + q * 1024;
            if (v >= N) continue;
            const bool dst = cur[q] == 0xFFFFFFFFu, src = r1[q] != r0[q], has_out = od[q] != 0;
            u32 f = (dst ? 1u : 0u) | (src ? 2u : 0u) | (has_out ? 4u : 0u);
            const u32 o = (v < nkl ? owner_hash_ref(ref_of_dense(v, nk, nl)) : owner_hash_obip(obi[q])) % d.world;   // = owner_of_dense(v)
            if (dst && has_out && o != d.rank) f |= (o + 1) << 3;
            fl[v] = (unsigned char)f;
        }
    }
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 lt = (1ull << lane) - 1ull;
    const u32 per_w = ((N + 1023) / 1024) * 64;                      // nodes per wave: a multiple of 64
    const u32 wb = wave * per_w < N ? wave * per_w : N, we = wb + per_w < N ? wb + per_w : N;
    constexpr int NL = REQ ? 10 : 2;                                 // (compile-time everywhere: a runtime bound would put the arrays into scratch)
    u32 cnt[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) cnt[j] = 0;
    for (u32 b = wb; b < we; b += 64) {                              // (uniform per wave)
        const u32 v = b + lane;
        const u32 f = v < we ? fl[v] : 0u;
        cnt[0] += (u32)__popcll(__ballot(((f & 2u) || ((f & 1u) && !(f & 4u))) ? 1 : 0));
        cnt[1] += (u32)__popcll(__ballot((f & 3u) ? 1 : 0));
        if (REQ) {
            const u32 o = f >> 3;
#pragma unroll
            for (int k = 0; k < 8; k++) cnt[(REQ ? 2 : 0) + (REQ ? k : 0)] += (u32)__popcll(__ballot(o == (u32)k + 1 ? 1 : 0));
        }
    }
    // wave totals -> LDS; thread j < NL turns list j's sixteen totals into exclusive prefixes (in place) and the list total
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NL; j++) wsum[j * 16 + wave] = cnt[j];
    }
    __syncthreads();
    if (threadIdx.x < (u32)NL) {
        u32 acc = 0;
        for (u32 w2 = 0; w2 < 16; w2++) { const u32 x = wsum[threadIdx.x * 16 + w2]; wsum[threadIdx.x * 16 + w2] = acc; acc += x; }
        wsum[160 + threadIdx.x] = acc;
    }
    __syncthreads();
    u32 run[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) run[j] = wsum[j * 16 + wave];
    if (threadIdx.x == 0) {
        d.ctr[C_ACT_L] = wsum[160]; d.ctr[C_ACT_P] = wsum[161];
        if (REQ) for (u32 k = 0; k < W; k++) {
            u32 t = wsum[162 + k];
            if (t > capp) { atomicAdd(&d.ctr[C_HALO_OVF], (u64)(t - capp)); t = capp; }
            req[(size_t)k * (capp + 1)] = t;
        }
    }
    for (u32 b = wb; b < we; b += 64) {
        const u32 v = b + lane;
        const u32 f = v < we ? fl[v] : 0u;
        {
            const bool in = (f & 2u) || ((f & 1u) && !(f & 4u));
            const u64 m = __ballot(in ? 1 : 0);
            if (in) d.act_l[run[0] + (u32)__popcll(m & lt)] = v;
            run[0] += (u32)__popcll(m);
        }
        {
            const bool in = (f & 3u) != 0;
            const u64 m = __ballot(in ? 1 : 0);
            if (in) d.act_p[run[1] + (u32)__popcll(m & lt)] = v;
            run[1] += (u32)__popcll(m);
        }
        if (REQ) {
            const u32 o = f >> 3;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                constexpr int J = REQ ? 2 : 0;
                const bool in = o == (u32)k + 1;
                const u64 m = __ballot(in ? 1 : 0);
                if (m) {                                             // (uniform)
                    const u32 pos = run[J + (REQ ? k : 0)] + (u32)__popcll(m & lt);
                    if (in && pos < capp) req[(size_t)k * (capp + 1) + 1 + pos] = v;
                    run[J + (REQ ? k : 0)] += (u32)__popcll(m);
                }
            }
        }
    }
}
__device__ __forceinline__ void build_lists(const Dev& d, u32* req, u32 capp, bool want_req, unsigned char* fl, u32* wsum) {
    const u32 N = (u32)d.ctr[C_N_NODES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const u32 W = d.world < 8 ? d.world : 8;
    if (N <= K6_FLAGS_LDS && !SG_ABL(d, 0x20000u)) { if (want_req) build_lists_staged<true>(d, req, capp, fl, wsum); else build_lists_staged<false>(d, req, capp, fl, wsum); return; }   // (uniform)
    const bool staged = N <= K6_FLAGS_LDS;
    if (staged) {
        for (u32 v0 = threadIdx.x; v0 < N; v0 += 4096) {             // four nodes per thread in flight
            u32 f[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const u32 v = v0 + q * 1024; f[q] = node_flags(d, v < N ? v : N - 1, nk, nl); }
#pragma unroll
            for (int q = 0; q < 4; q++) { const u32 v = v0 + q * 1024; if (v < N) fl[v] = (unsigned char)f[q]; }
        }
        __syncthreads();
    }
    const u32 per = (N + 1023) / 1024, beg = threadIdx.x * per < N ? threadIdx.x * per : N, end = beg + per < N ? beg + per : N;
    u32 c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cl = 0, cp = 0;
    for (u32 v = beg; v < end; v++) {
        const u32 f = staged ? fl[v] : node_flags(d, v, nk, nl);
        cl += ((f & 2u) || ((f & 1u) && !(f & 4u))) ? 1u : 0u;
        cp += (f & 3u) ? 1u : 0u;
        const u32 o = f >> 3;
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] += (o == (u32)k + 1);
    }
    // TEN exclusive block scans (two list cursors + eight owner cursors) in one go: wave scans of all ten values (DPP), the wave
    // totals through LDS, ONE barrier pair — ten block_excl_scan calls were thirty barriers and 42 us of a shard's window
    u32 val[10] = {cl, cp, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]}, pre[10], tot10[10];
    {
        const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32* ws = wsum;                                              // [10][16] wave totals (the caller provides >= 160 words)
#pragma unroll
        for (int j = 0; j < 10; j++) {
            u32 incl = val[j];
            incl += dpp32<0x111>(incl); incl += dpp32<0x112>(incl); incl += dpp32<0x114>(incl); incl += dpp32<0x118>(incl);   // row_shr 1, 2, 4, 8
            const u32 r0 = rdlane32(incl, 15), r1 = rdlane32(incl, 31), r2 = rdlane32(incl, 47);
            incl += (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
            pre[j] = incl - val[j];
            if (lane == 63) ws[j * 16 + wave] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 10; j++) {
            u32 before = 0, all = 0;
#pragma unroll
            for (u32 w2 = 0; w2 < 16; w2++) { const u32 x = ws[j * 16 + w2]; all += x; before += w2 < wave ? x : 0u; }
            pre[j] += before; tot10[j] = all;
        }
        __syncthreads();
    }
    u32 pl = pre[0], pp = pre[1];
    if (threadIdx.x == 0) { d.ctr[C_ACT_L] = tot10[0]; d.ctr[C_ACT_P] = tot10[1]; }
    u32 pos[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        pos[k] = pre[2 + k];
        if (want_req && (u32)k < W && threadIdx.x == 0) {
            u32 tot = tot10[2 + k];
            if (tot > capp) { atomicAdd(&d.ctr[C_HALO_OVF], (u64)(tot - capp)); tot = capp; }
            req[(size_t)k * (capp + 1)] = tot;
        }
    }
    for (u32 v = beg; v < end; v++) {
        const u32 f = staged ? fl[v] : node_flags(d, v, nk, nl);
        if ((f & 2u) || ((f & 1u) && !(f & 4u))) d.act_l[pl++] = v;
        if (f & 3u) d.act_p[pp++] = v;
        const u32 o = f >> 3;
        if (want_req && o) {
#pragma unroll
            for (int k = 0; k < 8; k++) if (o == (u32)k + 1) { if (pos[k] < capp) req[(size_t)k * (capp + 1) + 1 + pos[k]] = v; pos[k]++; }
        }
    }
}
// SMALL (host: ncap <= K6_FLAGS_LDS, so every window's N is): only the staged, wave-ordered builder is compiled in — the general
// form keeps ten-element arrays in scratch, and a kernel that one workgroup runs once per window pays for every cold
// instruction-cache line and for the scratch set-up.
template <bool SMALL>
__global__ __launch_bounds__(1024) void k6_active_lists(Dev d) {       // for the unpadded halo API
    __shared__ u32 wsum[176];
    __shared__ unsigned char fl[K6_FLAGS_LDS];
    if (SMALL) build_lists_staged<false>(d, nullptr, 0, fl, wsum); else build_lists(d, nullptr, 0, false, fl, wsum);
}

// ---- padded halo exchange (no host synchronisation: fixed-size all-to-all) ----------------------------
// req / serve layout: [world][capp + 1] u32, element 0 = count, ids follow.
template <bool SMALL>
__global__ __launch_bounds__(1024) void k6_halo_build_padded(Dev d, u32* req, u32 capp) {
    __shared__ u32 wsum[176];
    __shared__ unsigned char fl[K6_FLAGS_LDS];
    if (SMALL) build_lists_staged<true>(d, req, capp, fl, wsum); else build_lists(d, req, capp, true, fl, wsum);
}
// Round 4: the same lists by MANY workgroups in one launch (the one-workgroup builder above was 36 us of a C4 shard's window: a single
// CU walking every node).  Workgroup b owns the 1024 nodes from 1024 b, a thread per node: the node's flags (five loads, in flight
// together), the membership of the ten lists by wave ballots, the wave's counts through LDS; then the workgroup publishes its ten
// totals tagged with the launch epoch, sums those of the workgroups before it (they are resident: the grid is ncap / 1024 workgroups,
// dispatched in order — the assumption k2_rowptr makes) and writes its members at base + wave offset + rank among the lower lanes:
// ascending by construction, adjacent lanes write adjacent entries.  No reset, no second kernel, no flag array in memory.
#define K6M_LISTS 10
__global__ __launch_bounds__(1024) void k6_halo_lists(Dev d, u32* req, u32 capp, u32 epoch, u32 want_req) {
    const u32 N = (u32)d.ctr[C_N_NODES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const u32 W = d.world < 8 ? d.world : 8;
    const u32 b = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6, v = b * 1024u + t;
    if (b * 1024u >= N && b != 0) return;                            // beyond the last node (grid sized for ncap)
    __shared__ u32 wcnt[K6M_LISTS][16];
    __shared__ u32 base[K6M_LISTS], tot[K6M_LISTS];
    const u32 nkl = nk + nl;
    u32 f = 0;
    if (v < N) {
        const u32 cur = d.cursor[v], r0 = d.rowptr[v], r1 = d.rowptr[v + 1];
        const u64 od = d.st_sum[(size_t)v * SG_NODE_STAT_SUM_WORDS + ST_OUT_DEG];
        const u32 obi = v >= nkl ? d.ob_sorted[v - nkl] : 0u;
        const bool dst = cur == 0xFFFFFFFFu, src = r1 != r0, has_out = od != 0;
        f = (dst ? 1u : 0u) | (src ? 2u : 0u) | (has_out ? 4u : 0u);
        const u32 o = (v < nkl ? owner_hash_ref(ref_of_dense(v, nk, nl)) : owner_hash_obip(obi)) % d.world;   // = owner_of_dense(v)
        if (dst && has_out && o != d.rank) f |= (o + 1) << 3;
    }
    // list j: 0 = act_l (layer output computed here), 1 = act_p (score projections needed here), 2 + k = halo nodes owned by shard k
    const u64 lt = (1ull << lane) - 1ull;
    const bool in0 = (f & 2u) || ((f & 1u) && !(f & 4u)), in1 = (f & 3u) != 0;
    const u32 own1 = f >> 3;                                         // owner + 1 of a halo node, else 0
    const u64 m0 = __ballot(in0 ? 1 : 0), m1 = __ballot(in1 ? 1 : 0);
    u64 mk[8];
#pragma unroll
    for (int k = 0; k < 8; k++) mk[k] = want_req ? __ballot(own1 == (u32)k + 1 ? 1 : 0) : 0ull;
    if (lane == 0) {
        wcnt[0][wave] = (u32)__popcll(m0); wcnt[1][wave] = (u32)__popcll(m1);
#pragma unroll
        for (int k = 0; k < 8; k++) wcnt[2 + k][wave] = (u32)__popcll(mk[k]);
    }
    __syncthreads();
    if (t < K6M_LISTS) {                                             // thread j: list j's wave counts -> exclusive prefixes, the workgroup's total published,
        u32 acc = 0;                                                 // the totals of the workgroups before it summed
        for (u32 w2 = 0; w2 < 16; w2++) { const u32 x = wcnt[t][w2]; wcnt[t][w2] = acc; acc += x; }
        tot[t] = acc;
        __hip_atomic_store(&d.k6_tot[(size_t)b * 16 + t], ((u64)epoch << 32) | acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (relaxed: see k2_rowptr)
        u32 pre = 0;
        for (u32 j = 0; j < b; j++) {
            u64 x;
            do { x = __hip_atomic_load(&d.k6_tot[(size_t)j * 16 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((u32)(x >> 32) != epoch);
            pre += (u32)x;
        }
        base[t] = pre;
    }
    __syncthreads();
    if (in0) d.act_l[base[0] + wcnt[0][wave] + (u32)__popcll(m0 & lt)] = v;
    if (in1) d.act_p[base[1] + wcnt[1][wave] + (u32)__popcll(m1 & lt)] = v;
    if (want_req && own1) {
        const u32 k = own1 - 1;
        u64 m = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) m = k == (u32)q ? mk[q] : m;
        const u32 pos = base[2 + k] + wcnt[2 + k][wave] + (u32)__popcll(m & lt);
        if (pos < capp) req[(size_t)k * (capp + 1) + 1 + pos] = v;
    }
    if (t == 0 && (b + 1) * 1024u >= N) {                            // the workgroup of the last node knows the list lengths
        d.ctr[C_ACT_L] = base[0] + tot[0]; d.ctr[C_ACT_P] = base[1] + tot[1];
        if (want_req) for (u32 k = 0; k < W; k++) {
            u32 c = base[2 + k] + tot[2 + k];
            if (c > capp) { atomicAdd(&d.ctr[C_HALO_OVF], (u64)(c - capp)); c = capp; }
            req[(size_t)k * (capp + 1)] = c;
        }
    }
}
// rows[r][i][:] = feat[lists[r][1 + i]][:] for i < lists[r][0]   (pack: lists = what shard r asked of me)
__global__ __launch_bounds__(256) void k6_pack_padded(const float* __restrict__ feat, const u32* __restrict__ lists, u32 capp, u32 world, float* __restrict__ rows) {
    const u64 total = (u64)world * capp * 16;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < total; t += (u64)gridDim.x * 256) {
        const u32 q = (u32)(t & 15); const u64 ri = t >> 4; const u32 r = (u32)(ri / capp), i = (u32)(ri % capp);
        const u32* l = lists + (size_t)r * (capp + 1);
        if (i < l[0]) reinterpret_cast<float4*>(rows)[ri * 16 + q] = reinterpret_cast<const float4*>(feat)[(size_t)l[1 + i] * 16 + q];
    }
}
// feat[lists[r][1 + i]][:] = rows[r][i][:]   (unpack: lists = what I asked of shard r)
__global__ __launch_bounds__(256) void k6_unpack_padded(float* __restrict__ feat, const u32* __restrict__ lists, u32 capp, u32 world, const float* __restrict__ rows) {
    const u64 total = (u64)world * capp * 16;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < total; t += (u64)gridDim.x * 256) {
        const u32 q = (u32)(t & 15); const u64 ri = t >> 4; const u32 r = (u32)(ri / capp), i = (u32)(ri % capp);
        const u32* l = lists + (size_t)r * (capp + 1);
        if (i < l[0]) reinterpret_cast<float4*>(feat)[(size_t)l[1 + i] * 16 + q] = reinterpret_cast<const float4*>(rows)[ri * 16 + q];
    }
}
