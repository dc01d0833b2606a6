// sg_kernels.h — hand-written gfx950 kernels of the ServiceGraph engine (K1..K6).
// Included once by servicegraph.hip.  Every kernel is HBM/L2-bound integer or gather work except
// the per-node dense blocks of K4/K5, which run on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32).
#pragma once
#include "sg_device.h"

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ u32 sg_fmix32(u32 h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
__device__ __forceinline__ u32 hash_key64(u64 k) { return sg_fmix32((u32)k ^ sg_fmix32((u32)(k >> 32) + 0x9e3779b9u)); }

// owner shard of a node: by its stable ref; OBIP nodes (window-local indices) by their IP.
__host__ __device__ __forceinline__ u32 owner_hash_ref(u32 ref) { return sg_fmix32(ref); }
__host__ __device__ __forceinline__ u32 owner_hash_obip(u32 ip) { return sg_fmix32(ip ^ 0xA5A5F00Du); }

__host__ __device__ __forceinline__ u32 ip_h1(u32 ip, u32 bmask) { return sg_fmix32(ip) & bmask; }
__host__ __device__ __forceinline__ u32 ip_h2(u32 ip, u32 bmask) { return sg_fmix32(ip ^ 0x7F4A7C15u) & bmask; }

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
__device__ __forceinline__ bool ip_lookup(const u64* t, u32 mask, const u64* __restrict__ t2, u32 mask2, u32 ip, u32& pod, u32& svc) {
    const u64 e = ip_probe(t, mask, ip);
    if (e == SG_IP_EMPTY) return false;
    const u32 v = (u32)(e >> 32), kind = v >> 30, id = v & 0x3FFFFFFFu;
    if (kind == 1) pod = id;
    else if (kind == 2) svc = id;
    else { svc = id; const u64 e2 = ip_probe(t2, mask2, ip); if (e2 != SG_IP_EMPTY) pod = (u32)(e2 >> 32) & 0x3FFFFFFFu; }
    return true;
}

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
struct K1Local { u64 tmin, tmax; u32 maxlabel, dsrc, dcap, misr, acc; };
struct K1Ev { u64 key, dur, wt; u32 err; u32 alive; };
#define SG_DUR_MAX ((1ull << 62) - 1)      // durations saturate here: bits 62/63 of a single record carry flags

// the join: one event -> edge key, or a counted drop.
__device__ __forceinline__ bool k1_resolve(const Dev& d, const u64* iptab, const uint4 a, const uint4 b, K1Local& L, K1Ev& e) {
    const u32 saddr = a.x, daddr = a.y;
    const u32 status = a.w & 0xFFFFu, proto = (a.w >> 16) & 0xFFu, flags = a.w >> 24;
    const bool alive = (flags & SG_EV_ALIVE) != 0;                // an open connection, not a request (data.go:1628-1679)
    const u32 label = alive ? 0u : a.z;                          // ... joined without a Host header
    e.dur = (u64)b.x | ((u64)b.y << 32); e.wt = (u64)b.z | ((u64)b.w << 32);
    e.dur = e.dur > SG_DUR_MAX ? SG_DUR_MAX : e.dur;
    e.alive = alive ? 1u : 0u;

    u32 spod = SG_NONE, ssvc = SG_NONE;
    const bool sf = ip_lookup(iptab, d.ipmask, d.iptab2, d.ipmask2, saddr, spod, ssvc);
    if (!sf || spod == SG_NONE) { if (!alive) L.dsrc++; return false; }   // data.go:829-832: source must be a pod (:1643-1647 ignores silently)
    u32 from = SG_MAKE_REF(SG_REF_KNOWN, spod);
    const bool sharded = d.world > 1;
    u32 from_owner = sharded ? owner_hash_ref(from) : 0u;

    u32 dpod = SG_NONE, dsvc = SG_NONE, to, to_owner;
    const bool df = ip_lookup(iptab, d.ipmask, d.iptab2, d.ipmask2, daddr, dpod, dsvc);
    if (df && dsvc != SG_NONE) { to = SG_MAKE_REF(SG_REF_KNOWN, dsvc); to_owner = sharded ? owner_hash_ref(to) : 0u; }       // service first (:840-843)
    else if (df && dpod != SG_NONE) { to = SG_MAKE_REF(SG_REF_KNOWN, dpod); to_owner = sharded ? owner_hash_ref(to) : 0u; }  // then pod (:845-849)
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

// ---- variant 1: global open-addressing edge table + device-scope atomics.  General (any number
// of edges, any degree) but bound by the chip's ~22 G atomics/s and 12 ns per same-sector atomic
// (profiles/r01_atomic_probe.txt): kept for graphs the partitioned path cannot hold. ----------------
__global__ __launch_bounds__(256) void k1_resolve_aggregate(Dev d, const sg_event* __restrict__ ev, u64 n) {
    K1Local L; L.tmin = ~0ull; L.tmax = 0; L.maxlabel = L.dsrc = L.dcap = L.misr = L.acc = 0;
    const uint4* __restrict__ p = reinterpret_cast<const uint4*>(ev);
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint4 a = p[2 * i], b = p[2 * i + 1];
        K1Ev e;
        if (!k1_resolve(d, d.iptab, a, b, L, e)) continue;
        u32 slot;
        if (!table_slot(d.ekeys, d.emask, e.key, SG_EKEY_EMPTY, hash_key64(e.key) & d.emask, slot)) { if (!e.alive) { L.dcap++; L.acc--; } continue; }
        if (e.alive) continue;                                       // the slot exists now (count 0): that is all an open connection adds here
        u64* acc = d.eacc + (size_t)slot * 4;
        const u64 us = e.dur / 1000ull;
        atomicAdd(&acc[0], 1ull | ((u64)e.err << 32));
        atomicAdd(&acc[1], e.dur);
        atomicMax(&acc[2], e.dur);
        atomicAdd(&acc[3], us * us);
    }
    k1_publish_stats(d, L);
}

// ---- variant 0: partitioned aggregation, no device-scope atomics on the data path. ---------------
// Pass A (k1a_partition): each workgroup streams its contiguous share of the batch in chunks of
// 1024 events, aggregates a chunk in an LDS hash table (LDS atomics), then sweeps the table and
// appends one record per distinct edge to the slab piece [partition(edge)][this workgroup]:
// 16 bytes if the edge was hit once in the chunk, 40 bytes otherwise.  Hot edges collapse here, so
// partitions stay balanced; the long tail passes through as 16-byte singles.
// Pass B (k1b_merge, at window close): workgroup p owns partition p exclusively, merges its pieces
// in LDS and writes each distinct edge once with plain stores.
#define K1A_THREADS 1024
#define K1A_CT      2048      // LDS cache slots per workgroup
#define K1A_G       4         // events per thread per step
#define K1B_HT      1024
#define K1B_THREADS 1024
#define K1B_LPP     4        // lanes per piece
#define K1B_U       4

// Issue a global load NOW and leave it in flight; a later s_waitcnt (inline asm that names the
// destination registers as in/out operands) is the matching wait.  Written as inline asm because the
// compiler sinks a speculative load below the branch that makes its use conditional (k1b_merge:
// header -> test -> records became two dependent round trips) and puts waits between conditional
// loads.  vmcnt is in-order for loads, so the compiler's own (unaware) waits can only become
// stronger, never too weak.  Rule: no loop-carried value and no branch merge between an issue and
// its wait (a compiler-inserted register copy there would read a register that is still being loaded).
typedef u32 v4u_t __attribute__((ext_vector_type(4)));
typedef u32 v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gload16_issue(v4u_t& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload8_issue(v2u_t& dst, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory"); }

__device__ __forceinline__ u32 part_of_hash(const Dev& d, u32 hk) { return (hk >> 11) & (d.np - 1); }
__device__ __forceinline__ u32 part_of(const Dev& d, u64 key) { return part_of_hash(d, hash_key64(key)); }

__device__ __forceinline__ void ovf_append(const Dev& d, u64 key, u64 a0, u64 a1, u64 a2, u64 a3, K1Local& L) {
    const u64 idx = atomicAdd(&d.ctr[C_OVF_N], 1ull);
    if (idx < d.ovf_cap) { u64* o = d.ovf + idx * 5; o[0] = key; o[1] = a0; o[2] = a1; o[3] = a2; o[4] = a3; }
    else { const u32 c = (u32)(a0 & 0xFFFFFFFFull); L.dcap += c; L.acc -= c; }
}

// Slab geometry: piece (p, w) = records workgroup w produced for partition p, SG_PIECE_SLOTS 16-byte slots:
//   slot 0      header {n_single, n_aggregate, 0, 0}
//   slots 1..3  the piece's FIRST aggregate record {key, cnt | err<<32, sum_ns, max_ns, sumsq_us} (40 of 48 bytes)
//   slots 4..   ss single records {key.lo, key.hi, dur.lo, dur.hi | err << 31 | edge-only << 30}
//   slab_a[(p*nwg + w) * sa + r - 1] : the piece's aggregates r >= 1 (rare)
// Header, first aggregate and the first four singles share one 128-byte line; pass B reads a piece's header, first
// aggregate and first 16 singles in one round without touching slab_a.
#define SG_PIECE_HDR 4u
#define SG_PIECE_SLOTS(d) ((d).ss + SG_PIECE_HDR)
__device__ __forceinline__ uint4* piece_of(const Dev& d, u32 p, u32 w) { return d.slab_s + ((size_t)p * d.nwg + w) * SG_PIECE_SLOTS(d); }
// zero = 1: a record that only creates the edge (SG_EV_ALIVE): count 0, all accumulators 0
__device__ __forceinline__ void emit_single(const Dev& d, u32* fS, u32 w, u32 hk, u64 key, u64 dur, u32 err, K1Local& L, u32 zero = 0) {
    const u32 p = part_of_hash(d, hk);
    const u32 pos = atomicAdd(&fS[p], 1u);
    if (pos < d.ss) piece_of(d, p, w)[SG_PIECE_HDR + pos] = make_uint4((u32)key, (u32)(key >> 32), (u32)dur, (u32)(dur >> 32) | (err << 31) | (zero << 30));
    else if (zero) ovf_append(d, key, 0ull, 0ull, 0ull, 0ull, L);
    else { const u64 us = dur / 1000ull; ovf_append(d, key, 1ull | ((u64)err << 32), dur, dur, us * us, L); }
}
__device__ __forceinline__ void emit_agg(const Dev& d, u32* fA, u32 w, u64 key, u64 a0, u64 a1, u64 a2, u64 a3, K1Local& L) {
    const u32 p = part_of(d, key);
    const u32 pos = atomicAdd(&fA[p], 1u);
    if (pos < d.sa) {
        u64* o = pos == 0 ? reinterpret_cast<u64*>(piece_of(d, p, w) + 1) : d.slab_a + (((size_t)p * d.nwg + w) * d.sa + pos - 1) * 5;
        o[0] = key; o[1] = a0; o[2] = a1; o[3] = a2; o[4] = a3;
    }
    else ovf_append(d, key, a0, a1, a2, a3, L);
}

// Pass A.  One fat workgroup per CU streams a contiguous share of the batch with no barrier in the
// loop.  Each event is resolved (join table staged in LDS when IPLDS) and looked up in a
// first-come LDS cache (2 probes): the first key to claim a slot owns it for the whole launch and
// every later event of that key is folded in with LDS atomics; an event whose slots are taken by
// other keys bypasses the cache as a 16-byte single record.  Hot edges appear early, claim their
// slots and collapse to one aggregate record per workgroup; the long tail streams through.
template <bool IPLDS>
__global__ __launch_bounds__(K1A_THREADS) void k1a_partition(Dev d, const sg_event* __restrict__ ev, u64 n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* ckey = reinterpret_cast<u64*>(smem);                       // [K1A_CT]
    u64* cacc = ckey + K1A_CT;                                       // [K1A_CT][4]
    u32* fS = reinterpret_cast<u32*>(cacc + K1A_CT * 4);             // [np]
    u32* fA = fS + d.np;                                             // [np]
    u64* red = reinterpret_cast<u64*>(fA + d.np);                    // [8] workgroup statistics (WS_* order)
    u64* ipl = red + 8;                                              // [ipmask + 1] when IPLDS
    const u32 w = blockIdx.x, t = threadIdx.x;
    const uint4* __restrict__ pe = reinterpret_cast<const uint4*>(ev);
    const u64 per = (n + d.nwg - 1) / d.nwg;
    const u64 beg = (u64)w * per, end = (beg + per < n) ? beg + per : n;
    const bool first = d.batch_state == 1u;                          // first batch of the window: headers are zero by definition
    if (beg >= end) {                                                // no share of this batch: pieces and statistics stay as they are,
        if (first) for (u32 p = t; p < d.np; p += K1A_THREADS) *piece_of(d, p, w) = make_uint4(0, 0, 0, 0);   // but stale headers must go
        return;
    }
    const u64 last = end - 1;
    SG_STAMP(d, 0, 0);
    // K1A_G events per thread are fetched together (8 x 16 B in flight per lane).  Loads return in issue
    // order, so the small set-up loads (join table, this workgroup's piece headers) go out first, the
    // first event group right behind them; the LDS set-up then waits for the set-up loads only
    // (vmcnt(8): the eight event loads stay in flight).  Out-of-range lanes re-read the share's last
    // event and ignore it, so there is no branch between the loads.
    static_assert(K1A_G == 4 && SG_IP_LDS_MAX / K1A_THREADS == 4, "written out for 4 event pairs and 4 table words per lane");
    u64 i = beg + t;
#define K1A_ISSUE(base)                                                                                           \
        { const u64 j0 = (base), j1 = j0 + K1A_THREADS, j2 = j1 + K1A_THREADS, j3 = j2 + K1A_THREADS;               \
          const uint4* q0 = pe + 2 * (j0 < end ? j0 : last); const uint4* q1 = pe + 2 * (j1 < end ? j1 : last);     \
          const uint4* q2 = pe + 2 * (j2 < end ? j2 : last); const uint4* q3 = pe + 2 * (j3 < end ? j3 : last);     \
          gload16_issue(ea0, q0); gload16_issue(eb0, q0 + 1); gload16_issue(ea1, q1); gload16_issue(eb1, q1 + 1);   \
          gload16_issue(ea2, q2); gload16_issue(eb2, q2 + 1); gload16_issue(ea3, q3); gload16_issue(eb3, q3 + 1); }
    K1Local L; L.tmin = ~0ull; L.tmax = 0; L.maxlabel = L.dsrc = L.dcap = L.misr = L.acc = 0;
    const u64* iptab = IPLDS ? ipl : d.iptab;
    auto resolve = [&](const u64 idx, const v4u_t va, const v4u_t vb, K1Ev& e) -> bool {
        if (idx >= end) return false;
        const uint4 ca = make_uint4(va.x, va.y, va.z, va.w), cb = make_uint4(vb.x, vb.y, vb.z, vb.w);
        return k1_resolve(d, iptab, ca, cb, L, e);
    };
    auto insert = [&](const K1Ev& e) {
        const u32 hk = hash_key64(e.key);
        if (e.alive) { emit_single(d, fS, w, hk, e.key, 0ull, 0u, L, 1u); return; }
        u32 h = hk & (K1A_CT - 1);
        int slot = -1;
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
            u64 k = ((volatile u64*)ckey)[h];
            if (k == SG_EKEY_EMPTY) { k = atomicCAS(&ckey[h], SG_EKEY_EMPTY, e.key); if (k == SG_EKEY_EMPTY) k = e.key; }
            if (k == e.key) { slot = (int)h; break; }
            h = (h + 1) & (K1A_CT - 1);
        }
        if (slot >= 0) {
            const u64 us = e.dur / 1000ull;
            atomicAdd(&cacc[slot * 4], 1ull | ((u64)e.err << 32)); atomicAdd(&cacc[slot * 4 + 1], e.dur);
            atomicMax(&cacc[slot * 4 + 2], e.dur); atomicAdd(&cacc[slot * 4 + 3], us * us);
        } else emit_single(d, fS, w, hk, e.key, e.dur, e.err, L);
    };
    // One copy of the per-event code, run four times (not unrolled): the kernel body is executed once
    // per workgroup, so every instruction is an instruction-cache miss the first time through, and
    // four inlined copies of the join cost more in fetch stalls than the loop does in selects.
#define K1A_FOLD(base)                                                                                            \
        { asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea0), "+v"(eb0), "+v"(ea1), "+v"(eb1), "+v"(ea2), "+v"(eb2), "+v"(ea3), "+v"(eb3) : : "memory"); \
          _Pragma("unroll 1")                                                                                       \
          for (u32 q = 0; q < K1A_G; q++) {                                                                         \
              const v4u_t va = q == 0 ? ea0 : q == 1 ? ea1 : q == 2 ? ea2 : ea3;                                    \
              const v4u_t vb = q == 0 ? eb0 : q == 1 ? eb1 : q == 2 ? eb2 : eb3;                                    \
              K1Ev e;                                                                                               \
              if (resolve((base) + (u64)q * K1A_THREADS, va, vb, e)) insert(e);                                     \
          } }
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory")   /* LDS-only: does not drain the global stores */
    {
        // piece headers: zero by definition in the first batch of a window (no loads: 65 k scattered sectors per launch
        // saved); a later batch reads them with ordinary loads BEFORE anything is issued by hand — a hand-issued load
        // behind a branch would put a register merge between its issue and its wait (the compiler then copies the
        // still-loading register: seen in r02c, the LDS join table came out as garbage)
        for (u32 p = t; p < d.np; p += K1A_THREADS) { const uint4 h = first ? make_uint4(0, 0, 0, 0) : *piece_of(d, p, w); fS[p] = h.x; fA[p] = h.y; }
        v2u_t ip0, ip1, ip2, ip3;
        v4u_t ea0, eb0, ea1, eb1, ea2, eb2, ea3, eb3;
        if (IPLDS) {
            gload8_issue(ip0, d.iptab + (t & d.ipmask)); gload8_issue(ip1, d.iptab + ((t + K1A_THREADS) & d.ipmask));
            gload8_issue(ip2, d.iptab + ((t + 2 * K1A_THREADS) & d.ipmask)); gload8_issue(ip3, d.iptab + ((t + 3 * K1A_THREADS) & d.ipmask));
        }
        K1A_ISSUE(i);
        for (u32 k = t; k < K1A_CT; k += K1A_THREADS) ckey[k] = SG_EKEY_EMPTY;
        for (u32 k = t; k < K1A_CT * 4; k += K1A_THREADS) cacc[k] = 0;
        if (IPLDS) asm volatile("s_waitcnt vmcnt(8)" : "+v"(ip0), "+v"(ip1), "+v"(ip2), "+v"(ip3) : : "memory");
        if (IPLDS) {
            if (t <= d.ipmask) ipl[t] = (u64)ip0.x | ((u64)ip0.y << 32);
            if (t + K1A_THREADS <= d.ipmask) ipl[t + K1A_THREADS] = (u64)ip1.x | ((u64)ip1.y << 32);
            if (t + 2 * K1A_THREADS <= d.ipmask) ipl[t + 2 * K1A_THREADS] = (u64)ip2.x | ((u64)ip2.y << 32);
            if (t + 3 * K1A_THREADS <= d.ipmask) ipl[t + 3 * K1A_THREADS] = (u64)ip3.x | ((u64)ip3.y << 32);
        }
        if (t < 8) red[t] = t == WS_TMIN ? ~0ull : 0ull;
        LDS_BARRIER();
        SG_STAMP(d, 0, 1);
        K1A_FOLD(i);
        SG_STAMP(d, 0, 3);
    }
    for (i += (u64)K1A_G * K1A_THREADS; i < end; i += (u64)K1A_G * K1A_THREADS) {
        v4u_t ea0, eb0, ea1, eb1, ea2, eb2, ea3, eb3;
        K1A_ISSUE(i);
        K1A_FOLD(i);
    }
#undef K1A_ISSUE
#undef K1A_FOLD
    LDS_BARRIER();
    SG_STAMP(d, 0, 4);
    // flush the cache: one record per cached edge
    for (u32 s = t; s < K1A_CT; s += K1A_THREADS) {
        const u64 k = ckey[s];
        if (k == SG_EKEY_EMPTY) continue;
        const u64 x0 = cacc[s * 4], x1 = cacc[s * 4 + 1], x2 = cacc[s * 4 + 2], x3 = cacc[s * 4 + 3];
        if ((x0 & 0xFFFFFFFFull) == 1ull) emit_single(d, fS, w, hash_key64(k), k, x1, (u32)(x0 >> 32), L);
        else emit_agg(d, fA, w, k, x0, x1, x2, x3, L);
    }
    // workgroup statistics: wave reduce -> LDS -> one thread updates this workgroup's private line
    {
        const u64 tmin = wave_min_u64(L.tmin), tmax = wave_max_u64(L.tmax);
        const u32 ml = (u32)wave_max_u64(L.maxlabel);
        const u32 ds = wave_sum_u32(L.dsrc), dc = wave_sum_u32(L.dcap), mr = wave_sum_u32(L.misr), ac = wave_sum_u32(L.acc);
        if ((t & 63) == 0) {
            if (ac) { atomicMin(&red[WS_TMIN], tmin); atomicMax(&red[WS_TMAX], tmax); atomicAdd(&red[WS_ACCEPTED], (u64)ac); }
            if (ml) atomicMax(&red[WS_MAXLABEL], (u64)ml);
            if (ds) atomicAdd(&red[WS_DROPPED_SRC], (u64)ds);
            if (dc) atomicAdd(&red[WS_DROPPED_CAP], (u64)dc);
            if (mr) atomicAdd(&red[WS_MISROUTED], (u64)mr);
        }
    }
    LDS_BARRIER();
    for (u32 p = t; p < d.np; p += K1A_THREADS)
        *piece_of(d, p, w) = make_uint4(fS[p] < d.ss ? fS[p] : d.ss, fA[p] < d.sa ? fA[p] : d.sa, 0u, 0u);
    SG_STAMP(d, 0, 5);
    if (t == 0) {
        u64* g = d.wgstat + (size_t)(blockIdx.x % SG_MAX_K1_WGS) * WS_WORDS;
        if (red[WS_ACCEPTED]) { atomicMin(&g[WS_TMIN], red[WS_TMIN]); atomicMax(&g[WS_TMAX], red[WS_TMAX]); atomicAdd(&g[WS_ACCEPTED], red[WS_ACCEPTED]); }
        if (red[WS_MAXLABEL]) atomicMax(&g[WS_MAXLABEL], red[WS_MAXLABEL]);
        if (red[WS_DROPPED_SRC]) atomicAdd(&g[WS_DROPPED_SRC], red[WS_DROPPED_SRC]);
        if (red[WS_DROPPED_CAP]) atomicAdd(&g[WS_DROPPED_CAP], red[WS_DROPPED_CAP]);
        if (red[WS_MISROUTED]) atomicAdd(&g[WS_MISROUTED], red[WS_MISROUTED]);
    }
    SG_STAMP(d, 0, 6);
#undef LDS_BARRIER
}

// Pass B.  Workgroup p owns partition p: it reads its nwg pieces, merges them in an LDS table and
// writes every distinct edge once with plain stores:
//   e_from/e_to [p*pcap + i]  dense endpoints        acc_src [(p*pcap + i)*4]  accumulators
//   deg[from] += 1 (atomic u32; a row's edges are spread over the partitions; the returned value is the
//   edge's position inside its CSR row)
__global__ __launch_bounds__(K1B_THREADS) void k1b_merge(Dev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* hkey = reinterpret_cast<u64*>(smem);                       // [K1B_HT]
    u64* hacc = hkey + K1B_HT;                                       // [K1B_HT][4]
    __shared__ u32 n_drop, out_n;
    const u32 p = blockIdx.x, t = threadIdx.x;
    SG_STAMP(d, 1, 0);
    // counters the tail needs: fetched now so their latency hides behind the merge
    const u64 ovf_n = d.ctr[C_OVF_N];
    const u32 nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS], nob = (u32)d.ctr[C_N_OBIP];
    static_assert(K1B_U == 4, "the speculative first round is written out for 4 singles per lane");
    const u32 sub = t % K1B_LPP;
    // K1B_LPP lanes walk one piece (record r belongs to lane r % K1B_LPP).  A piece's header, the lane's
    // first K1B_U singles and its first aggregate are fetched together (the slab memory is always
    // mapped, stale contents are ignored): one round of latency for a typical piece (<= 16 singles)
    // instead of three dependent ones.  The first round is in flight during the LDS set-up.
#define K1B_SIDX(u) (SG_PIECE_HDR + ((sub + (u) * K1B_LPP) < d.ss ? (sub + (u) * K1B_LPP) : 0))
#define K1B_ISSUE(piece)                                                                                  \
        gload16_issue(hv, (piece));                                                                        \
        gload16_issue(xv0, (piece) + K1B_SIDX(0)); gload16_issue(xv1, (piece) + K1B_SIDX(1));              \
        gload16_issue(xv2, (piece) + K1B_SIDX(2)); gload16_issue(xv3, (piece) + K1B_SIDX(3));              \
        gload16_issue(y01, (piece) + 1); gload16_issue(y23, (piece) + 2); gload8_issue(y4, (piece) + 3)
#define K1B_WAIT() asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv), "+v"(xv0), "+v"(xv1), "+v"(xv2), "+v"(xv3), "+v"(y01), "+v"(y23), "+v"(y4) : : "memory")
    const u32 w0 = t / K1B_LPP, wc = w0 < d.nwg ? w0 : 0;
    const uint4* piece0 = piece_of(d, p, wc);
    const u64* pa0 = d.slab_a + ((size_t)p * d.nwg + wc) * d.sa * 5;
    v4u_t hv, xv0, xv1, xv2, xv3, y01, y23; v2u_t y4;
    K1B_ISSUE(piece0);
    for (u32 i = t; i < K1B_HT; i += K1B_THREADS) hkey[i] = SG_EKEY_EMPTY;
    for (u32 i = t; i < K1B_HT * 4; i += K1B_THREADS) hacc[i] = 0;
    if (t == 0) { n_drop = 0; out_n = 0; }
    __syncthreads();

    auto add = [&](u64 key, u64 a0, u64 a1, u64 a2, u64 a3) {
        u32 h = hash_key64(key) & (K1B_HT - 1); bool ok = false;
        for (u32 it = 0; it < K1B_HT; it++) {                        // bounded: the table holds at most K1B_HT distinct edges
            u64 k = ((volatile u64*)hkey)[h];
            if (k == SG_EKEY_EMPTY) { k = atomicCAS(&hkey[h], SG_EKEY_EMPTY, key); if (k == SG_EKEY_EMPTY) k = key; }
            if (k == key) { ok = true; break; }
            h = (h + 1) & (K1B_HT - 1);
        }
        if (!ok) { atomicAdd(&n_drop, (u32)(a0 & 0xFFFFFFFFull)); return; }
        atomicAdd(&hacc[h * 4], a0); atomicAdd(&hacc[h * 4 + 1], a1); atomicMax(&hacc[h * 4 + 2], a2); atomicAdd(&hacc[h * 4 + 3], a3);
    };
    auto merge_piece = [&](const uint4* piece, const u64* pa, const v4u_t h, const v4u_t q0, const v4u_t q1, const v4u_t q2, const v4u_t q3,
                           const v4u_t z01, const v4u_t z23, const v2u_t z4) {
        if (!(h.x | h.y)) return;
        uint4 x[K1B_U] = {make_uint4(q0.x, q0.y, q0.z, q0.w), make_uint4(q1.x, q1.y, q1.z, q1.w),
                          make_uint4(q2.x, q2.y, q2.z, q2.w), make_uint4(q3.x, q3.y, q3.z, q3.w)};
        u64 y[5] = {(u64)z01.x | ((u64)z01.y << 32), (u64)z01.z | ((u64)z01.w << 32), (u64)z23.x | ((u64)z23.y << 32),
                    (u64)z23.z | ((u64)z23.w << 32), (u64)z4.x | ((u64)z4.y << 32)};
        const u32 ns = h.x < d.ss ? h.x : d.ss, na = h.y < d.sa ? h.y : d.sa;
        for (u32 r0 = sub; r0 < ns; r0 += K1B_LPP * K1B_U) {
            if (r0 != sub) {
#pragma unroll
                for (int u = 0; u < K1B_U; u++) if (r0 + u * K1B_LPP < ns) x[u] = piece[SG_PIECE_HDR + r0 + u * K1B_LPP];
            }
#pragma unroll
            for (int u = 0; u < K1B_U; u++) if (r0 + u * K1B_LPP < ns) {
                const u64 key = (u64)x[u].x | ((u64)x[u].y << 32), dur = (u64)x[u].z | ((u64)(x[u].w & 0x3FFFFFFFu) << 32), us = dur / 1000ull;
                const u64 one = ((x[u].w >> 30) & 1u) ? 0ull : 1ull;             // bit 62: edge-only record (SG_EV_ALIVE)
                add(key, one | ((u64)(x[u].w >> 31) << 32), dur, dur, us * us);
            }
        }
        for (u32 r0 = sub; r0 < na; r0 += K1B_LPP) {                 // aggregate 0 came with the piece; the others (rare) from slab_a
            if (r0 != 0) { const u64* q = pa + (size_t)(r0 - 1) * 5; y[0] = q[0]; y[1] = q[1]; y[2] = q[2]; y[3] = q[3]; y[4] = q[4]; }
            add(y[0], y[1], y[2], y[3], y[4]);
        }
    };
    SG_STAMP(d, 1, 1);
    K1B_WAIT();
    SG_STAMP(d, 1, 2);
    const bool empty = d.batch_state == 2u;                          // no batch this window: the pieces are the previous window's
    if (w0 < d.nwg && !empty) merge_piece(piece0, pa0, hv, xv0, xv1, xv2, xv3, y01, y23, y4);
    for (u32 w = w0 + K1B_THREADS / K1B_LPP; w < d.nwg && !empty; w += K1B_THREADS / K1B_LPP) {    // only when nwg > 256
        const uint4* piece = piece_of(d, p, w);
        const u64* pa = d.slab_a + ((size_t)p * d.nwg + w) * d.sa * 5;
        v4u_t hv, xv0, xv1, xv2, xv3, y01, y23; v2u_t y4;
        K1B_ISSUE(piece);
        K1B_WAIT();
        merge_piece(piece, pa, hv, xv0, xv1, xv2, xv3, y01, y23, y4);
    }
#undef K1B_ISSUE
#undef K1B_WAIT
#undef K1B_SIDX
    __syncthreads();
    SG_STAMP(d, 1, 3);
    // (no reset of the piece headers: the first batch of the next window rewrites every one of them)
    {
        const u64 no = ovf_n < d.ovf_cap ? ovf_n : d.ovf_cap;
        for (u64 i = t; i < no; i += K1B_THREADS) {
            const u64* o = d.ovf + i * 5;
            if (part_of(d, o[0]) == p) add(o[0], o[1], o[2], o[3], o[4]);
        }
    }
    __syncthreads();
    SG_STAMP(d, 1, 4);

    // compact the table into the partition's output slots (order within a partition is arbitrary;
    // the CSR row sort makes the final order canonical)
#pragma unroll
    for (u32 q = 0; q < K1B_HT / K1B_THREADS; q++) {
        const u32 s = q * K1B_THREADS + t;
        const u64 k = hkey[s];
        if (k == SG_EKEY_EMPTY) continue;
        const u32 f = dense_of(d, (u32)(k >> 32), nk, nl, nob), to = dense_of(d, (u32)k, nk, nl, nob);
        if (f == SG_NONE || to == SG_NONE) { atomicAdd(&n_drop, (u32)(hacc[s * 4] & 0xFFFFFFFFull)); continue; }
        const u32 i = atomicAdd(&out_n, 1u);
        if (i >= d.pcap) { atomicAdd(&n_drop, (u32)(hacc[s * 4] & 0xFFFFFFFFull)); continue; }
        const size_t slot = (size_t)p * d.pcap + i;
        d.e_from[slot] = f; d.e_to[slot] = to;
        ulonglong2* o = reinterpret_cast<ulonglong2*>(d.acc_src + slot * 4);
        o[0] = make_ulonglong2(hacc[s * 4], hacc[s * 4 + 1]); o[1] = make_ulonglong2(hacc[s * 4 + 2], hacc[s * 4 + 3]);
        d.e_rank[slot] = atomicAdd(&d.deg[SG_DEG_IDX(f, p & (SG_DEG_REP - 1))], 1u);        // arrival order inside the row's replica
    }
    __syncthreads();
    SG_STAMP(d, 1, 5);
    if (t == 0) {
        d.part_n[p] = out_n < d.pcap ? out_n : d.pcap;
        if (n_drop) atomicAdd(&d.ctr[C_DROPPED_CAP], (u64)n_drop);
    }
}

// ------------------------------------------------------------------------------------------------
// K2  csr_build: canonical node numbering, CSR with sorted rows.
// ------------------------------------------------------------------------------------------------
// one workgroup (1024 threads): window bookkeeping.
//   (a) fold the per-workgroup K1 statistics into the counters and re-arm the slots;
//   (b) collect the window's distinct raw outbound IPs (or take the sharded driver's union list),
//       sort them (bitonic, global memory), drop duplicates -> ob_sorted, N_OBIP;
//   (c) N = NK + NL + NOB.
__global__ __launch_bounds__(1024) void kc_prepare(Dev d, u64 n_known, u64 n_labels_decl, u32* list, const u32* n_in, u32 list_cap, u32 collect,
                                                   const u32* seg, u32 seg_stride, u32 seg_world) {
    __shared__ u64 red[7][16];
    __shared__ u32 wsum[17];
    __shared__ u32 cnt;
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
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
            u64 nl = d.ctr[C_N_LABELS];                              // labels are cumulative across windows
            nl = red[2][0] > nl ? red[2][0] : nl; nl = n_labels_decl > nl ? n_labels_decl : nl;
            d.ctr[C_N_LABELS] = nl; d.ctr[C_N_KNOWN] = n_known;
            d.ctr[C_DROPPED_SRC] = red[3][0]; d.ctr[C_MISROUTED] = red[5][0]; d.ctr[C_N_EVENTS] = red[6][0];
            d.ctr[C_DROPPED_CAP] = red[4][0];                        // K1b / K2 add their own drops afterwards
            d.ctr[C_N_LONG] = 0;                                     // k2_rowptr's workgroups append to the long-row list
        }
    }
    // (b)
    u32 n;
    if (collect == 1) {
        for (u32 i = t; i <= d.obmask; i += 1024) {
            const u64 k = d.obkeys[i];
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
        d.ctr[C_N_NODES] = d.ctr[C_N_KNOWN] + d.ctr[C_N_LABELS] + nob;   // thread 0 wrote both above
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
// wait for it — all workgroups are resident, at most one per CU, so the wait cannot deadlock).
__global__ __launch_bounds__(1024) void k2_rowptr(Dev d, u32 epoch) {
    const u32 N = (u32)d.ctr[C_N_NODES];
    __shared__ u32 wsum[17];
    __shared__ u32 rdeg[K2_RP_ROWS];
    __shared__ u32 nlong, lbase, pre;
    const u32 b = blockIdx.x, t = threadIdx.x, r0 = b * K2_RP_ROWS;
    if (r0 >= N && b != 0) return;                                   // beyond the last row (grid sized for ncap)
    if (t == 0) nlong = 0;
    // 1. replicas -> in-row offsets, row degrees
    for (u32 pass = 0; pass < K2_RP_ROWS / 128; pass++) {
        const u32 rl = pass * 128 + (t >> 3), row = r0 + rl, rep = t & 7;
        u32 v = row < N ? d.deg[SG_DEG_IDX(row, rep)] : 0u;
        u32 incl = v;                                                // inclusive prefix over the 8 lanes of the row
#pragma unroll
        for (int s2 = 1; s2 < 8; s2 <<= 1) { const u32 o = __shfl_up(incl, s2, 8); if ((int)rep >= s2) incl += o; }
        if (row < N) d.deg[SG_DEG_IDX(row, rep)] = incl - v;
        if (rep == 7) rdeg[rl] = incl;
    }
    __syncthreads();
    // 2. local scan
    const u32 dg = t < K2_RP_ROWS ? rdeg[t] : 0u;
    u32 total;
    const u32 run = block_excl_scan<1024>(dg, wsum, &total);
    // 3. totals of the preceding workgroups
    if (t == 0) {
        __atomic_store_n(&d.rp_tot[b], ((u64)epoch << 32) | total, __ATOMIC_RELEASE);
        pre = 0;
    }
    __syncthreads();
    {
        u32 mine = 0;
        for (u32 j = t; j < b; j += 1024) {
            u64 x;
            do { x = __atomic_load_n(&d.rp_tot[j], __ATOMIC_ACQUIRE); } while ((u32)(x >> 32) != epoch);
            mine += (u32)x;
        }
        if (b) { mine = wave_sum_u32(mine); if ((t & 63) == 0 && mine) atomicAdd(&pre, mine); }
    }
    __syncthreads();
    const u32 base = pre;
    // 4. publish
    if (t < K2_RP_ROWS && r0 + t < N) {
        d.rowptr[r0 + t] = base + run;
        if (dg > 64) atomicAdd(&nlong, 1u);
    }
    __syncthreads();
    if (t == 0) lbase = nlong ? (u32)atomicAdd(&d.ctr[C_N_LONG], (u64)nlong) : 0u;   // C_N_LONG is zeroed by kc_prepare
    __syncthreads();
    {   // positions inside this workgroup's slice of the long-row list
        __shared__ u32 lpos;
        if (t == 0) lpos = 0;
        __syncthreads();
        if (t < K2_RP_ROWS && r0 + t < N && dg > 64) d.longrows[lbase + atomicAdd(&lpos, 1u)] = r0 + t;
    }
    if (t == 0) {
        if (b == 0) {
            d.ctr[C_OVF_N] = 0;                                        // K1b has consumed the overflow list
            d.ctr[C_ACT_L] = SG_ACT_NONE; d.ctr[C_ACT_P] = 0;          // no active lists yet for this window (see k6_active_lists)
        }
        if (r0 + K2_RP_ROWS >= N) {                                    // the workgroup of the last row knows E
            const u32 E = base + total;
            d.rowptr[N] = E;
            d.ctr[C_N_EDGES] = (u64)E < d.max_edges ? E : d.max_edges;
            if (d.variant == 0) { d.ctr[C_EDGES_FOUND] = E; if ((u64)E > d.max_edges) d.ctr[C_DROPPED_CAP] += (u64)E - d.max_edges; }
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
            continue;
        }
        const u32 pos = d.rowptr[f] + atomicAdd(&d.cursor[f], 1u);
        d.col[pos] = d.e_to[i]; d.cslot[pos] = d.e_slot[i];
    }
}
__global__ __launch_bounds__(256) void k2_scatter_parts(Dev d) {
    const u32 p = blockIdx.x, n = d.part_n[p];
    for (u32 i = threadIdx.x; i < n; i += 256) {
        const u32 slot = p * d.pcap + i;
        const u32 f = d.e_from[slot];
        const u64 pos = (u64)d.rowptr[f] + d.deg[SG_DEG_IDX(f, p & (SG_DEG_REP - 1))] + d.e_rank[slot];   // row + replica offset + arrival order
        if (pos < d.max_edges) { d.col[pos] = d.e_to[slot]; d.cslot[pos] = slot; }
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
struct EdgeEmitArgs { u64* acc_csr; u32* csr_from; u64* eacc; u64* ekeys; u32* alive_csr; u32 variant; };
__device__ __forceinline__ void edge_emit(const EdgeEmitArgs d, u32 pos, u32 row, u32 slot, u64, u64, u64, const ulonglong2 x, const ulonglong2 y) {
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(d.acc_csr + (size_t)pos * 4);
    dst[0] = x; dst[1] = y;
    d.csr_from[pos] = row;
    d.alive_csr[pos] = 0;                                            // k3_in_stats adds the window's open connections
    if (d.variant == 1) {                                            // variant 1: this is also the window reset of the edge table
        ulonglong2* src = reinterpret_cast<ulonglong2*>(d.eacc + (size_t)slot * 4);
        src[0] = make_ulonglong2(0, 0); src[1] = make_ulonglong2(0, 0);
        d.ekeys[slot] = SG_EKEY_EMPTY;
    }
}
// e_uv, lat_z, err_ratio of edge `pos` from its accumulators and its row's out-statistics
__device__ __forceinline__ void edge_features(const Dev& d, u32 pos) {
    const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)pos * 4);
    const ulonglong2 x = a[0], y = a[1];
    const u64* rs = d.st_sum + (size_t)d.csr_from[pos] * SG_NODE_STAT_SUM_WORDS;
    const u64 r_cnt = rs[ST_OUT_CNT], r_sum = rs[ST_OUT_SUM], r_ssq = rs[ST_OUT_SSQ];
    const u64 cnt = x.x & 0xFFFFFFFFull, err = x.x >> 32, sum = x.y, mx = y.x, ssq = y.y;
    const double m_e = mean_us(sum, cnt), s_e = std_us(sum, ssq, cnt);
    const double mu = mean_us(r_sum, r_cnt), sd = std_us(r_sum, r_ssq, r_cnt);
    const double z = (m_e - mu) / (sd > 1.0 ? sd : 1.0);
    const float lat_z = (float)z;
    const float err_ratio = cnt ? (float)((double)err / (double)cnt) : 0.0f;
    const float zc = lat_z < -8.0f ? -8.0f : (lat_z > 8.0f ? 8.0f : lat_z);
    float4* e = reinterpret_cast<float4*>(d.efeat + (size_t)pos * SG_F_EDGE);
    e[0] = make_float4((float)log1p((double)cnt), (float)log1p(m_e / 1000.0), (float)log1p(s_e / 1000.0), (float)log1p((double)mx / 1e6));
    e[1] = make_float4(err_ratio, (float)log1p((double)err), zc * 0.125f, 1.0f);
    d.latz[pos] = lat_z; d.errr[pos] = err_ratio;
}

#define K2_SORT_LDS 4096
#define K2_LONG_WGS 1024
__global__ __launch_bounds__(256) void k2_rowsort_gather(Dev d) {
    const u32 N = (u32)d.ctr[C_N_NODES], nlong = (u32)d.ctr[C_N_LONG];
    const EdgeEmitArgs ea = {d.acc_csr, d.csr_from, d.eacc, d.ekeys, d.alive_csr, d.variant};
    __shared__ u32 sk[K2_SORT_LDS], sv[K2_SORT_LDS];
    __shared__ u64 red[5][4];
    __shared__ u32 bsum[5];
    const u32 BW = (N + 31) >> 5;                                    // words of a node bitmap
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- long rows: the first K2_LONG_WGS workgroups, one row at a time each ----
    const u32 nlw = gridDim.x > 2 * K2_LONG_WGS ? K2_LONG_WGS : (gridDim.x / 2 ? gridDim.x / 2 : 1);   // host launches >= 2 workgroups
    if (blockIdx.x < nlw) {
        for (u32 li = blockIdx.x; li < nlong; li += nlw) {
            const u32 rr = d.longrows[li];
            const u32 b = d.rowptr[rr];
            u32 m = d.rowptr[rr + 1] - b;
            if ((u64)b + m > d.max_edges) m = b < d.max_edges ? (u32)(d.max_edges - b) : 0;
            if (m == 0) continue;
            u32* key = d.col + b; u32* val = d.cslot + b;
            u64 cnt = 0, err = 0, sum = 0, ssq = 0, mx = 0;
            if (BW <= K2_SORT_LDS) {
                // Bitmap rank: the destinations of one row are distinct node ids < N, so setting bit `to` in an
                // N-bit LDS bitmap and counting the bits below it IS the sorted position — O(m + N/32) per row
                // instead of a comparison sort (a 3000-edge hub row cost ~100 us in the bitonic network).
                for (u32 w = threadIdx.x; w < BW; w += 256) sk[w] = 0;
                __syncthreads();
                for (u32 i = threadIdx.x; i < m; i += 256) { const u32 k = key[i]; atomicOr(&sk[k >> 5], 1u << (k & 31)); }
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
                u32* gk = d.sort_k + 2 * (size_t)b; u32* gv = d.sort_v + 2 * (size_t)b;     // the row's private slice of the scratch
                for (u32 i = threadIdx.x; i < m; i += 256) {
                    const u32 k = key[i], v = val[i];
                    const u32 r = sv[k >> 5] + __popc(sk[k >> 5] & ((1u << (k & 31)) - 1u));
                    gk[r] = k; gv[r] = v;
                }
                __threadfence_block();
                __syncthreads();
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
            } else if (m <= 1024) {
                // rank sort: keys in LDS, every thread counts the smaller keys of its (<= 4) elements
                for (u32 i = threadIdx.x; i < m; i += 256) sk[i] = key[i];
                __syncthreads();
                u32 mk[4], mv[4], rk[4]; ulonglong2 ax[4], ay[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32 i = threadIdx.x + q * 256;
                    mk[q] = i < m ? sk[i] : 0xFFFFFFFFu; mv[q] = i < m ? val[i] : 0u; rk[q] = 0;
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
                if (m > K2_SORT_LDS) { gk = d.sort_k + 2 * (size_t)b; gv = d.sort_v + 2 * (size_t)b; }   // private padded slice of the global scratch
                for (u32 i = threadIdx.x; i < np2; i += 256) { gk[i] = i < m ? key[i] : 0xFFFFFFFFu; gv[i] = i < m ? val[i] : 0; }
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
        const u32 k = lane < n ? d.col[beg + lane] : 0xFFFFFFFFu, v = lane < n ? d.cslot[beg + lane] : 0;
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
        }
        if (lane < n) edge_emit(ea, beg + rank, r, v, cnt, sum, ssq, x, y);
    }
}

// ---- in-statistics: per destination node, reduce over its in-edges --------------------------------
// A handful of workgroups each accumulate a contiguous range of CSR positions in LDS (node-indexed
// arrays when ncap <= K3_IN_NODES, else a hash table refilled per round) with LDS atomics, then
// flush the touched nodes with device-scope atomics: a popular service with thousands of in-edges
// receives one atomic per word per workgroup instead of one per edge.
#define K3_IN_NODES 2560
#define K3_IN_HT    2048
#define K3_IN_ROUND 8192      // edges per round of the hashed path
#define K3_IN_PROBES 32
#define K3_IN_WGS   16
__device__ __forceinline__ void in_flush(const Dev& d, u32 to, const u64* o) {
    u64* gsum = d.st_sum + (size_t)to * SG_NODE_STAT_SUM_WORDS;
    atomicAdd(&gsum[ST_IN_DEG], o[0]); atomicAdd(&gsum[ST_IN_CNT], o[1]); atomicAdd(&gsum[ST_IN_ERR], o[2]);
    atomicAdd(&gsum[ST_IN_SUM], o[3]); atomicAdd(&gsum[ST_IN_SSQ], o[4]);
    atomicMax(&d.st_max[(size_t)to * 2 + 1], o[5]);
}
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

__global__ __launch_bounds__(1024) void k3_in_stats(Dev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 E = (u32)d.ctr[C_N_EDGES], N = (u32)d.ctr[C_N_NODES];
    const u32 G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
    alive_mark(d, g, G, t);
    const u32 per = (E + G - 1) / G, p0 = g * per < E ? g * per : E, p1 = p0 + per < E ? p0 + per : E;
    if (d.in_dense) {
        u64* acc = reinterpret_cast<u64*>(smem);                     // [N][6]: deg, cnt, err, sum, ssq, max
        for (u32 i = t; i < N * 6; i += 1024) acc[i] = 0;
        __syncthreads();
        for (u32 pb = p0 + t; pb < p1; pb += 1024 * 4) {          // 4 edges per thread in flight
            u32 to[4]; ulonglong2 x[4], y[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 p = pb + q * 1024;
                if (p < p1) { to[q] = d.col[p]; const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)p * 4); x[q] = a[0]; y[q] = a[1]; }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) if (pb + q * 1024 < p1) {
                u64* o = acc + (size_t)to[q] * 6;
                atomicAdd(&o[0], 1ull); atomicAdd(&o[1], x[q].x & 0xFFFFFFFFull); atomicAdd(&o[2], x[q].x >> 32);
                atomicAdd(&o[3], x[q].y); atomicAdd(&o[4], y[q].y); atomicMax(&o[5], y[q].x);
            }
        }
        __syncthreads();
        for (u32 v = t; v < N; v += 1024) if (acc[(size_t)v * 6]) in_flush(d, v, acc + (size_t)v * 6);
    } else {
        u64* tacc = reinterpret_cast<u64*>(smem);                    // [K3_IN_HT][6]
        u32* tkey = reinterpret_cast<u32*>(tacc + K3_IN_HT * 6);     // [K3_IN_HT]
        // Rounds of K3_IN_ROUND edges (8 per thread, loads of a thread's edges in flight together).  The table holds
        // K3_IN_HT distinct destinations; a destination that finds no slot within K3_IN_PROBES probes (a round with
        // too many distinct destinations) is added with device-scope atomics directly — rare, and exact either way.
        for (u32 c0 = p0; c0 < p1; c0 += K3_IN_ROUND) {
            for (u32 i = t; i < K3_IN_HT; i += 1024) tkey[i] = SG_NONE;
            for (u32 i = t; i < K3_IN_HT * 6; i += 1024) tacc[i] = 0;
            __syncthreads();
            const u32 c1 = c0 + K3_IN_ROUND < p1 ? c0 + K3_IN_ROUND : p1;
            for (u32 pb = c0 + t; pb < c1; pb += 1024 * 4) {
                u32 to[4]; ulonglong2 x[4], y[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32 p = pb + q * 1024;
                    if (p < c1) { to[q] = d.col[p]; const ulonglong2* a = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)p * 4); x[q] = a[0]; y[q] = a[1]; }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) if (pb + q * 1024 < c1) {
                    u32 h = sg_fmix32(to[q]) & (K3_IN_HT - 1);
                    bool found = false;
                    for (u32 pr = 0; pr < K3_IN_PROBES; pr++) {
                        u32 kk = ((volatile u32*)tkey)[h];
                        if (kk == SG_NONE) { kk = atomicCAS(&tkey[h], SG_NONE, to[q]); if (kk == SG_NONE) kk = to[q]; }
                        if (kk == to[q]) { found = true; break; }
                        h = (h + 1) & (K3_IN_HT - 1);
                    }
                    if (found) {
                        u64* o = tacc + (size_t)h * 6;
                        atomicAdd(&o[0], 1ull); atomicAdd(&o[1], x[q].x & 0xFFFFFFFFull); atomicAdd(&o[2], x[q].x >> 32);
                        atomicAdd(&o[3], x[q].y); atomicAdd(&o[4], y[q].y); atomicMax(&o[5], y[q].x);
                    } else {
                        const u64 one[6] = {1ull, x[q].x & 0xFFFFFFFFull, x[q].x >> 32, x[q].y, y[q].y, y[q].x};
                        in_flush(d, to[q], one);
                    }
                }
            }
            __syncthreads();
            for (u32 s = t; s < K3_IN_HT; s += 1024) if (tkey[s] != SG_NONE) in_flush(d, tkey[s], tacc + (size_t)s * 6);
            __syncthreads();
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
    for (u64 i = tid; i < nc * SG_DEG_REP; i += nt) d.deg[i * SG_DEG_STRIDE] = 0;
        for (u64 i = tid; i < nc; i += nt) d.cursor[i] = 0;
    for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_SUM_WORDS; i += nt) d.st_sum[i] = 0;
    for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_MAX_WORDS; i += nt) d.st_max[i] = 0;
    for (u64 i = tid; i <= d.obmask; i += nt) d.obkeys[i] = 0;
    if (d.variant == 1 && d.ctr[C_EDGES_FOUND] > d.max_edges) {
        for (u64 i = tid; i <= d.emask; i += nt) {
            d.ekeys[i] = SG_EKEY_EMPTY;
            ulonglong2* a = reinterpret_cast<ulonglong2*>(d.eacc + (size_t)i * 4);
            a[0] = make_ulonglong2(0, 0); a[1] = make_ulonglong2(0, 0);
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
#define SG_MEAN_BLOCK 512        // neighbours per block of the canonical mean: block sums are added in block order

// 16-node tiles, 1024 threads: in the gather phase every wave owns one node of the tile (the rows
// follow a power law, so per-node parallelism is what bounds this kernel); the dense phase runs on
// the first 4 waves.  With PROJ the tile's fresh h rows are immediately projected to the score
// head's P = b1 + h Wu and Q = h Wv (last layer, unsharded), saving a launch.
#define K4_HUB_BLOCKS 32         // block sums of a hub row kept in LDS per round
// NT = 1024 (one wave per tile row) for the 32-feature first layer; NT = 512 (a wave takes two rows) for the
// 64-feature hidden layers: twice the registers per lane, so all 16 loads of a batch in the 16-byte layout are in
// flight (a quarter of the round trips on hub rows; C3 layer 2: 369 us before).
template <int FI, bool USE_MFMA, bool PROJ, int NT>
__global__ __launch_bounds__(NT) void k4_sage_layer(Dev d, const float* __restrict__ hin, float* __restrict__ hout, const float* __restrict__ Wl, const float* __restrict__ Wh) {
    constexpr int LDA = 2 * FI + 2;                               // +2 floats: conflict-free A-fragment reads
    constexpr int LDH = SG_F_HID + 2;
    __shared__ float A[16 * LDA];
    __shared__ float H[PROJ ? 16 * LDH : 1];
    __shared__ u32 skip[16];
    __shared__ u32 vid[16], tdeg[16];
    __shared__ float hub[K4_HUB_BLOCKS * FI];
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
                // block 0 here (one wave per row, all rows at once); the further blocks of a hub row below
                if (deg) gather_block_sum<FI, WIDE>(hin, d.col + beg, 0, deg < SG_MEAN_BLOCK ? deg : SG_MEAN_BLOCK, row + FI);
                if (lane < FI) {                                     // (same wave wrote row[FI..): ordered by the LDS counter)
                    const float t = deg ? row[FI + lane] : 0.0f;
                    row[FI + lane] = deg > SG_MEAN_BLOCK ? t : (deg ? t / (float)deg : 0.0f);
                }
            }
            if (lane == 0) { skip[r] = sk ? 1u : 0u; vid[r] = v; tdeg[r] = deg; }
        }
        __syncthreads();
        // hub rows (more than one block): the blocks of a row are spread over the 16 waves, the block sums are
        // then added in block order by one wave — a 3000-neighbour row no longer serialises on a single wave
        for (u32 r = 0; r < 16; r++) {
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
                    for (int r = 0; r < 4; r++) { const u32 row = (lane >> 4) * 4 + r; if (v0 + row < N) dst[(size_t)(v0 + row) * SG_F_HID + jb + i] = c[r]; }
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
                        for (int c = 0; c < 4; c++) dst[(size_t)(v0 + row) * SG_F_HID + jq + c] = acc[c];
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
                if (v0 + row < N) { d.P[(size_t)vid[row] * SG_F_HID + jb + i] = p[r]; d.Q[(size_t)vid[row] * SG_F_HID + jb + i] = q[r]; }
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
                for (int c = 0; c < 4; c++) { d.P[(size_t)vid[row] * SG_F_HID + jq + c] = p[c]; d.Q[(size_t)vid[row] * SG_F_HID + jq + c] = q[c]; }
        }
        __syncthreads();
    }
}

// one wave scores K5_U edges at a time (lane = hidden unit j): the index loads, then the 2*K5_U row
// gathers of a step are independent and in flight together.
#define K5_U 4
template <bool RESET>
__global__ __launch_bounds__(256) void k5_edge_score(Dev d, const float* __restrict__ Wh) {
    const u32 E = (u32)d.ctr[C_N_EDGES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const float* __restrict__ We = Wh + 2 * SG_F_HID * SG_F_HID;
    const float* __restrict__ w2 = We + SG_F_EDGE * SG_F_HID + SG_F_HID;
    const float b2 = w2[SG_F_HID];
    const u32 lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    float we[SG_F_EDGE];
#pragma unroll
    for (int k = 0; k < (int)SG_F_EDGE; k++) we[k] = We[k * SG_F_HID + lane];
    const float w2j = w2[lane];
    for (u32 p0 = wave * K5_U; p0 < E; p0 += nw * K5_U) {
        u32 uu[K5_U], vv[K5_U]; float t[K5_U], ev[K5_U];
#pragma unroll
        for (int q = 0; q < K5_U; q++) { const u32 p = p0 + q < E ? p0 + q : E - 1; uu[q] = d.csr_from[p]; vv[q] = d.col[p]; }
        // what the row writer (lane q -> edge p0 + q) needs is fetched now, beside the index loads, not after the sums
        const u32 pw = p0 + (lane < K5_U ? lane : 0) < E ? p0 + (lane < K5_U ? lane : 0) : E - 1;
        const ulonglong2* __restrict__ aw = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)pw * 4);
        const ulonglong2 wx = aw[0], wy = aw[1];
        const float w_latz = d.latz[pw], w_errr = d.errr[pw]; const u32 w_alive = d.alive_csr[pw];
#pragma unroll
        for (int q = 0; q < K5_U; q++) {
            const u32 p = p0 + q < E ? p0 + q : E - 1;
            t[q] = d.P[(size_t)uu[q] * SG_F_HID + lane] + d.Q[(size_t)vv[q] * SG_F_HID + lane];
            ev[q] = lane < SG_F_EDGE ? d.efeat[(size_t)p * SG_F_EDGE + lane] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < K5_U; q++) {
            float x = t[q];
#pragma unroll
            for (int k = 0; k < (int)SG_F_EDGE; k++) x = fmaf(__uint_as_float(rdlane32(__float_as_uint(ev[q]), k)), we[k], x);
            x = x > 0.0f ? x : 0.0f;
            float r = x * w2j;
            t[q] = wave_butterfly_sum_f32(r);
        }
        if (lane < K5_U && p0 + lane < E) {
            const u32 p = p0 + lane;
            float r = t[0];
#pragma unroll
            for (int q = 1; q < K5_U; q++) r = lane == (u32)q ? t[q] : r;
            u32 u = uu[0], v = vv[0];
#pragma unroll
            for (int q = 1; q < K5_U; q++) { u = lane == (u32)q ? uu[q] : u; v = lane == (u32)q ? vv[q] : v; }
            const float logit = r + b2;
            const float score = 1.0f / (1.0f + expf(-logit));
            sg_edge_out o;
            o.sum_ns = wx.y; o.max_ns = wy.x; o.sumsq_us = wy.y;
            o.from_ref = ref_of_dense(u, nk, nl); o.to_ref = ref_of_dense(v, nk, nl);
            o.count = (u32)(wx.x & 0xFFFFFFFFull); o.err_count = (u32)(wx.x >> 32);
            o.score = score; o.lat_z = w_latz; o.err_ratio = w_errr; o.alive = w_alive;
            d.rows[p] = o;
        }
    }
    if (RESET) {
        // Window reset folded into the last kernel of the pipeline (nothing after it reads these arrays;
        // the counters stay: sg_window_read / the next kc_prepare consume them).
        const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, nt = (u64)gridDim.x * 256;
        const u64 nc = (u64)d.ncap + 1;
        for (u64 i = tid; i < nc * SG_DEG_REP; i += nt) d.deg[i * SG_DEG_STRIDE] = 0;
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
__device__ __forceinline__ void build_lists(const Dev& d, u32* req, u32 capp, bool want_req, unsigned char* fl, u32* wsum) {
    const u32 N = (u32)d.ctr[C_N_NODES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const u32 W = d.world < 8 ? d.world : 8;
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
    u32 tl, tp;
    u32 pl = block_excl_scan<1024>(cl, wsum, &tl);
    __syncthreads();
    u32 pp = block_excl_scan<1024>(cp, wsum, &tp);
    __syncthreads();
    if (threadIdx.x == 0) { d.ctr[C_ACT_L] = tl; d.ctr[C_ACT_P] = tp; }
    u32 pos[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        pos[k] = 0;
        if (want_req && (u32)k < W) {                                // uniform
            u32 tot;
            pos[k] = block_excl_scan<1024>(c[k], wsum, &tot);
            if (threadIdx.x == 0) {
                if (tot > capp) { atomicAdd(&d.ctr[C_HALO_OVF], (u64)(tot - capp)); tot = capp; }
                req[(size_t)k * (capp + 1)] = tot;
            }
            __syncthreads();
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
__global__ __launch_bounds__(1024) void k6_active_lists(Dev d) {       // for the unpadded halo API
    __shared__ u32 wsum[17];
    __shared__ unsigned char fl[K6_FLAGS_LDS];
    build_lists(d, nullptr, 0, false, fl, wsum);
}

// ---- padded halo exchange (no host synchronisation: fixed-size all-to-all) ----------------------------
// req / serve layout: [world][capp + 1] u32, element 0 = count, ids follow.
__global__ __launch_bounds__(1024) void k6_halo_build_padded(Dev d, u32* req, u32 capp) {
    __shared__ u32 wsum[17];
    __shared__ unsigned char fl[K6_FLAGS_LDS];
    build_lists(d, req, capp, true, fl, wsum);
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
