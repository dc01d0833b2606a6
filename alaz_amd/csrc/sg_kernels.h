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

#include "sg_k1_wide.h"     // K1 with 16-byte records (small windows, the histogram, k1_variant = 2)
#include "sg_k1_narrow.h"   // the narrow-record form of both passes (default of variant 0)
#include "sg_k1_team.h"     // round 4: pass A with two teams per workgroup and a batched join
#include "sg_k2.h"          // K2: kc_prepare, the rebuild chain, the row sort
#include "sg_kw.h"          // warm / delta windows: kw_capture, kw_compact
#include "sg_k3.h"          // K3: in-statistics, node + edge features, the window reset
#include "sg_k4.h"          // K4: gather-mean + dense (MFMA)
#include "sg_k5.h"          // K5: projections + edge score
#include "sg_k6.h"          // K6: halo lists, pack / unpack
