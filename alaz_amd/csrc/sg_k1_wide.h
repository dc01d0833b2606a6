// sg_k1_wide.h — K1 with 16-byte records (k1a_partition + k1b_merge): small windows (BASELINE config 2), the per-edge histogram, k1_variant = 2
// Part of the kernel translation unit: included by sg_kernels.h (which holds the shared helpers), in this order.
#pragma once

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
