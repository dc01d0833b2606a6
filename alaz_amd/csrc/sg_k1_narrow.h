// sg_k1_narrow.h — K1 resolve_aggregate, narrow-record form (the default of variant 0).  Included by sg_kernels.h.
//
// Same job as k1a_partition / k1b_merge (extractAddressPair + setFromToV2 + ReverseDirection + PersistRequest,
// aggregator/data.go:1760-1767, 827-870; datastore/dto.go:226-231; backend.go:819-847), different exchange format:
//
//   * an edge is the pair of COMPACT endpoint indices (cf, ct) < 2^nb; sg_kmix (sg_hash.h) is a bijection of that pair, its
//     top pb bits are the partition, the other rb = 2 nb - pb <= 31 bits ("rem") name the edge inside the partition.  A
//     record is therefore 8 bytes {duration u32, rem | error << 31} instead of 16, and pass B keys its LDS table by a u32;
//   * pass A sorts every tile of 8192 records by partition in LDS before it writes them: the records of a partition leave as
//     one contiguous run written by adjacent lanes (full or nearly full 64/128-byte write requests) instead of one scattered
//     16-byte store per record.  The r02 kernel issued 8.6 M write requests per C3 window (one per record,
//     profiles/r02_store_probe.txt: the cost is per request, not per byte); a run of 16 records is ~3.
//
// Everything that does not fit a narrow record (durations of 2^32 ns and more, SG_EV_ALIVE edge-only records) travels as a
// 16-byte "wide" single in its own region of the piece; hot keys are folded in the first-come LDS cache exactly as before
// and leave as 40-byte aggregates.  Integer adds / max only: bit-exact whatever the order.
#pragma once

#define K1T_THREADS 1024
#ifndef K1B_WARM_U
#define K1B_WARM_U 2            // pass B of an engine that keeps state: 16-byte loads per lane in the rolling buffer (two workgroups per CU: 64 registers)
#endif
#define K1T_CHUNK   4096u         // events per chunk at most: four per thread
#define K1T_TS(NSUB) (K1T_CHUNK * (NSUB))   // records per tile: NSUB chunks (1 or 2)
#define K1T_NONE    0xFFFFFFFFu
#define K1T_RANK_SHIFT 12         // stash word: partition (<= 12 bits) | rank in the partition's run << 12

__device__ __forceinline__ u32 ci_of_ref(const Dev& d, u32 ref) {
    const u32 t = SG_REF_TYPE(ref), v = SG_REF_VALUE(ref);
    return t == SG_REF_KNOWN ? v : (t == SG_REF_LABEL ? d.max_known + v : d.max_known + d.max_labels + v);
}
__device__ __forceinline__ u32 ref_of_ci(const Dev& d, u32 c) {
    if (c < d.max_known) return SG_MAKE_REF(SG_REF_KNOWN, c);
    if (c < d.max_known + d.max_labels) return SG_MAKE_REF(SG_REF_LABEL, c - d.max_known);
    return SG_MAKE_REF(SG_REF_OBIP, c - d.max_known - d.max_labels);
}
__device__ __forceinline__ u64* piece8(const Dev& d, u32 p, u32 w) { return d.slab8 + ((size_t)p * d.nwg + w) * d.punits; }

// a record that found no room in its piece: the window's overflow list (pass B filters it by partition); mk = mixed key
__device__ __forceinline__ void ovf8_single(const Dev& d, u32 p, u64 mk, u64 dur, u32 err, u32 zero, K1Local& L) {
    if (zero) ovf_append(d, p, mk, 0ull, 0ull, 0ull, 0ull, L);
    else { const u64 us = dur / 1000ull; ovf_append(d, p, mk, 1ull | ((u64)err << 32), dur, dur, us * us, L); }
}
// one narrow record outside the tile (cache flush): position from the piece's LDS counter
__device__ __forceinline__ void emit_narrow_direct(const Dev& d, u32* fcn, u32 w, u32 p, u32 rem, u32 dur, u32 err, K1Local& L) {
    const u32 pos = atomicAdd(&fcn[p], 1u);
    if (pos < d.sn) piece8(d, p, w)[pos] = (u64)dur | ((u64)(rem | (err << 31)) << 32);
    else { atomicSub(&fcn[p], 1u); ovf8_single(d, p, ((u64)p << d.rb) | rem, (u64)dur, err, 0u, L); }
}
// a wide single: {mixed key, dur | err << 63 | edge-only << 62}; fcw[p] = wide singles | aggregates << 16
__device__ __forceinline__ void emit_wide(const Dev& d, u32* fcw, u32 w, u32 p, u64 mk, u64 dur, u32 err, u32 zero, K1Local& L) {
    const u32 pos = atomicAdd(&fcw[p], 1u) & 0xFFFFu;
    if (pos < d.sw) reinterpret_cast<uint4*>(piece8(d, p, w) + d.sn)[pos] = make_uint4((u32)mk, (u32)(mk >> 32), (u32)dur, (u32)(dur >> 32) | (err << 31) | (zero << 30));
    else { atomicSub(&fcw[p], 1u); ovf8_single(d, p, mk, dur, err, zero, L); }
}
// one aggregate of this launch's cache; an aggregate of the same key left by an EARLIER launch of the window is updated
// in place (the piece is private to this workgroup and a key is flushed by exactly one lane: no atomics needed)
__device__ __forceinline__ void emit_agg8(const Dev& d, u32* fcw, u32 w, u32 p, u64 mk, u64 a0, u64 a1, u64 a2, u64 a3, K1Local& L, bool first) {
    u64* ag = piece8(d, p, w) + d.sn + 2 * d.sw;
    if (!first) {
        u32 na0 = d.hdr8[(size_t)p * d.nwg + w].y >> 16; na0 = na0 < d.sa ? na0 : d.sa;
        for (u32 r = 0; r < na0; r++) {
            u64* o = ag + 5 * r;
            if (o[0] != mk) continue;
            o[1] += a0; o[2] += a1; { const u64 m = o[3]; o[3] = a2 > m ? a2 : m; } o[4] += a3;
            return;
        }
    }
    const u32 pos = atomicAdd(&fcw[p], 1u << 16) >> 16;
    if (pos < d.sa) { u64* o = ag + 5 * pos; o[0] = mk; o[1] = a0; o[2] = a1; o[3] = a2; o[4] = a3; }
    else { atomicSub(&fcw[p], 1u << 16); ovf_append(d, p, mk, a0, a1, a2, a3, L); }
}

// ---- pass A ---------------------------------------------------------------------------------------------------------
// 256 workgroups x 1024 threads, one per CU; one launch per ingested batch.  LDS: edge cache | piece counters | run counters
// (two sets, alternating tiles) | run offsets | statistics | tile | join tables.
// Per tile (8 events per thread, fetched as two groups of four):
//   P1  join + setFromToV2 as selects + key mix + cache probe (the r02 fast path); a record that is not folded into the
//       cache takes a rank in its partition's run (returning LDS add) and stays in registers
//   P2  every wave turns the run lengths into offsets       P3  every thread drops its records at offset + rank
//   P4  thread i copies tile position i to its piece: adjacent lanes, adjacent addresses (at the top of the NEXT tile, behind its first loads)
// Two LDS-only barriers per tile (behind P1 and P3); the run counters alternate so that P4 of tile k may overlap P1 of tile k + 1.
// L2M: level 2 of the join 0 = read from global memory, 1 = staged in LDS as it is (u32 entries), 2 = staged in LDS as u16
// entries kind << 14 | id (engines whose id space fits 14 bits: half the LDS, which goes to the edge cache).
// NSUB: chunks per tile (2: 8192-record tiles = longer runs, 24 registers of parked records; 1: 4096-record tiles = half the
// LDS, which goes to the edge cache, half the registers, twice the barriers per event).
template <int L2M, bool SHARDED, int NSUB>
__global__ __launch_bounds__(K1T_THREADS) void k1a_tile_partition(Dev d, const sg_event* __restrict__ ev, u64 n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 CT = d.k1a_ct, NP = d.np;
    u64* ckey = reinterpret_cast<u64*>(smem);                       // [CT] mixed keys
    u64* cacc = ckey + CT;                                           // [CT][4]
    u32* fcn2 = reinterpret_cast<u32*>(cacc + (size_t)CT * 4);       // [2][np] narrow records in piece (p, this workgroup): the set a tile's copy-out reads,
                                                                     //         and the set its scan writes for the next tile
    u32* fcw = fcn2 + 2 * NP;                                        // [np] wide singles | aggregates << 16
    u32* bcnt = fcw + NP;                                            // [2][np] run lengths of the tile being built / written
    u32* boff = bcnt + 2 * NP;                                       // [np] run offsets inside the tile
    u64* red = reinterpret_cast<u64*>(boff + NP);                    // [8] workgroup statistics (WS_* order)
    u64* tile = red + 8;                                             // [K1T_TS(NSUB)]
    uint4* jl = reinterpret_cast<uint4*>(tile + K1T_TS(NSUB));       // LDS copy of the join blob: jl1 | jl2 (L2M != 0)
    const u64* l1 = reinterpret_cast<const u64*>(jl);
    const u32* l2 = L2M == 1 ? reinterpret_cast<const u32*>(l1 + d.jl1mask + 1) : d.jl2;
    const unsigned short* l2h = reinterpret_cast<const unsigned short*>(l1 + d.jl1mask + 1);      // L2M == 2
    const u32 w = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint4* __restrict__ pe = reinterpret_cast<const uint4*>(ev);
    // The batch is cut into chunks of `chunk` events (1024 .. 4096, a multiple of 1024: up to four events per lane and group);
    // chunk c belongs to workgroup c % nwg.  At any moment the 256 workgroups therefore read ONE contiguous stretch of the batch
    // (256 x 128 KB) instead of 256 streams 1.25 MB apart (which lean on a few HBM channels at a time), and a small batch
    // (n <= 1024 nwg) still gives every workgroup its 1024 events, so the pieces of a window fed by many small batches fill evenly.
    const u64 per = (n + d.nwg - 1) / d.nwg;
    const u32 chunk = per >= 4096 ? 4096u : (u32)((per + 1023) / 1024 * 1024);
    const u64 nchunk = (n + chunk - 1) / chunk;
    const u64 end = n;
    const bool first = d.batch_state == 1u;                          // first batch of the window: the headers are zero by definition
    // chunk c belongs to workgroup (c + rot) % nwg: the host advances `rot` by the chunks of every launch, so that a window fed by many
    // SMALL batches (a launch of 4 096 events has four chunks) does not put all of its records into the pieces of workgroups 0..3
    // (round 4: a paced stream of 4 096-event batches into a 2 048-partition engine overflowed exactly those pieces: a third of the
    // events counted as dropped)
    const u32 wr = (w + d.nwg - d.k1a_rot % d.nwg) % d.nwg;
    if ((u64)wr >= nchunk) {                                         // no share of this batch: pieces and statistics stay as they are,
        if (first) for (u32 p = t; p < NP; p += K1T_THREADS) d.hdr8[(size_t)p * d.nwg + w] = make_uint2(0u, 0u);   // but stale headers must go
        return;
    }
    SG_STAMP(d, 0, 0);
    // the shader clock this launch ran at (sg_clock_probe): cycles / 100 MHz ticks of workgroup 0, thread 0
    const bool clk_me = w == 0 && t == 0;
    const u64 clk_c0 = clk_me ? __builtin_readcyclecounter() : 0ull, clk_r0 = clk_me ? wall_clock64() : 0ull;
    // statistics: accepted events and their time-stamp range stay in registers; everything rare (drops, labels, misrouted events, what
    // the general path and the overflow paths count) is added to the workgroup's LDS line where it happens — six registers less
    // across the fold
    u32 st_acc = 0; u64 st_tmin = ~0ull, st_tmax = 0;
    auto lflush = [&](const K1Local& x) {
        if (x.acc) { atomicAdd(&red[WS_ACCEPTED], (u64)x.acc); atomicMin(&red[WS_TMIN], x.tmin); atomicMax(&red[WS_TMAX], x.tmax); }
        if (x.lost) atomicAdd(&red[WS_PAD], (u64)x.lost);
        if (x.maxlabel) atomicMax(&red[WS_MAXLABEL], (u64)x.maxlabel);
        if (x.dsrc) atomicAdd(&red[WS_DROPPED_SRC], (u64)x.dsrc);
        if (x.dcap) atomicAdd(&red[WS_DROPPED_CAP], (u64)x.dcap);
        if (x.misr) atomicAdd(&red[WS_MISROUTED], (u64)x.misr);
    };
#define K1T_LNEW(L) K1Local L; L.tmin = ~0ull; L.tmax = 0; L.maxlabel = L.dsrc = L.dcap = L.misr = L.acc = L.lost = 0
    const u32 nb = d.nb, nbmask = (1u << nb) - 1u, pshift = nb - d.pb, rbmask = (1u << d.rb) - 1u, bmask = CT / 2 - 1;
    const bool ck_any = d.ck_n != 0;

    // events (base) + k * 1024, k = 0..3, below `cend`; out-of-range lanes re-read the chunk's first event and ignore it
#define K1T_ISSUE(base, cend, cfirst)                                                                             \
        { const u64 j0 = (base), j1 = j0 + K1T_THREADS, j2 = j1 + K1T_THREADS, j3 = j2 + K1T_THREADS;               \
          const uint4* q0 = pe + 2 * (j0 < (cend) ? j0 : (cfirst)); const uint4* q1 = pe + 2 * (j1 < (cend) ? j1 : (cfirst)); \
          const uint4* q2 = pe + 2 * (j2 < (cend) ? j2 : (cfirst)); const uint4* q3 = pe + 2 * (j3 < (cend) ? j3 : (cfirst)); \
          gload16_issue(ea0, q0); gload16_issue(eb0, q0 + 1); gload16_issue(ea1, q1); gload16_issue(eb1, q1 + 1);   \
          gload16_issue(ea2, q2); gload16_issue(eb2, q2 + 1); gload16_issue(ea3, q3); gload16_issue(eb3, q3 + 1); }
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory")   /* LDS-only: does not drain the global stores */
    // cache fold of one accepted event whose bucket the caller has read (k0, k1); returns false when the event must travel
    auto cache_fold = [&](u32 bucket, u64 mk, u64 k0, u64 k1, u64 dur, u32 err) -> bool {
        int slot = k0 == mk ? (int)(2u * bucket) : (k1 == mk ? (int)(2u * bucket + 1u) : -1);
        if (slot < 0 && (k0 == SG_EKEY_EMPTY || k1 == SG_EKEY_EMPTY)) slot = cache_claim(ckey, bucket, mk, k0, k1);
        if (slot < 0) return false;
        u64 ssq;
        if ((dur >> 32) == 0) { const u32 us = div1000_u32((u32)dur); ssq = (u64)us * (u64)us; }
        else { const u64 us = dur / 1000ull; ssq = us * us; }
        atomicAdd(&cacc[slot * 4], 1ull | ((u64)err << 32)); atomicAdd(&cacc[slot * 4 + 1], dur);
        atomicMax(&cacc[slot * 4 + 2], dur); atomicAdd(&cacc[slot * 4 + 3], ssq);
        return true;
    };
    // The general path (rare events: open connections, raw-IP outbound destinations, IPs in both maps or in the residual
    // cuckoo table, durations of 2^32 ns and more, labels out of range): the full join on the global tables.  A request
    // that fits a narrow record comes back as one (slo, shi, spr) and joins the tile like a fast-path record.
    auto general = [&](const v4u_t va, const v4u_t vb, u32* bc, u32& slo, u32& shi, u32& spr) {
        spr = K1T_NONE;
        K1Ev e;
        K1T_LNEW(L);
        const bool ok = k1_resolve(d, make_uint4(va.x, va.y, va.z, va.w), make_uint4(vb.x, vb.y, vb.z, vb.w), L, e);
        if (!ok) { lflush(L); return; }
        u32 Lm, Rm;
        sg_kmix(ci_of_ref(d, (u32)(e.key >> 32)), ci_of_ref(d, (u32)e.key), nbmask, &Lm, &Rm);
        const u32 part = Lm >> pshift;
        const u64 mk = ((u64)Lm << nb) | Rm;
        if (e.alive) { emit_wide(d, fcw, w, part, mk, 0ull, 0u, 1u, L); lflush(L); return; }
        const u32 bkt = Rm & bmask;
        if (cache_fold(bkt, mk, lds_fresh_u64(&ckey[2 * bkt]), lds_fresh_u64(&ckey[2 * bkt + 1]), e.dur, e.err)) { lflush(L); return; }
        if (e.dur >> 32) { emit_wide(d, fcw, w, part, mk, e.dur, e.err, 0u, L); lflush(L); return; }
        const u32 rank = atomicAdd(&bc[part], 1u);
        slo = (u32)e.dur; shi = ((u32)mk & rbmask) | (e.err << 31); spr = part | (rank << K1T_RANK_SHIFT);
        lflush(L);
    };
    constexpr u32 KSH = L2M == 2 ? 14u : 30u, IDM = L2M == 2 ? 0x3FFFu : 0x3FFFFFFFu;   // kind shift / id mask of a level-2 entry as this build reads it
    auto join = [&](u32 ip) -> u32 {
        const u32 b = ip >> 8;
        const u64 e1 = l1[((__umul24(b, SG_JL1_K1)) >> 9) & d.jl1mask], e2 = l1[((__umul24(b, SG_JL1_K2)) >> 11) & d.jl1mask];
        const u32 blk = (u32)e1 == b ? (u32)(e1 >> 32) : ((u32)e2 == b ? (u32)(e2 >> 32) : 0u);     // block 0 = the all-zero block
        return L2M == 2 ? (u32)l2h[(blk << 8) | (ip & 255u)] : l2[(blk << 8) | (ip & 255u)];
    };
    // The fast path, one event: branch-free join (two-level block table in LDS), data.go:827-870 as selects, key mix,
    // read-only cache probe.  `rare` hands the event to the general path instead.
    auto fast = [&](const u64 idx, const u64 cend, const v4u_t va, const v4u_t vb, u32* bc, bool& rare_out, u32& slo, u32& shi, u32& spr) {
        const bool inr = idx < cend;
        const u32 flags = va.w >> 24, label = va.z;
        const u32 vs = join(va.x), vd = join(va.y);
        const u32 ks = vs >> KSH, kd = vd >> KSH;
        bool rare = ((flags & SG_EV_ALIVE) != 0) | (ks == 3u) | (kd == 3u) | (vb.y != 0u) |
                    ((kd == 0u) & ((label == 0u) | (label > d.max_labels))) | (ck_any & ((vs == 0u) | (vd == 0u)));
        rare &= inr; rare_out = rare;
        const bool fastv = inr & !rare;
        bool acc = fastv & (ks == 1u);                               /* data.go:829-832: the source must be a pod */
        if (fastv & (ks != 1u)) atomicAdd(&red[WS_DROPPED_SRC], 1ull);
        u32 cf = vs & IDM;
        u32 ct = kd ? (vd & IDM) : (d.max_known + label - 1u);           /* service / pod id, else Host label (:840-854) */
        if (acc & (kd == 0u)) atomicMax(&red[WS_MAXLABEL], (u64)label);
        if (flags & SG_EV_REVERSE) { const u32 x_ = cf; cf = ct; ct = x_; }      /* dto.go:226-231 */
        if (SHARDED) { const bool mine = (owner_hash_ref(ref_of_ci(d, cf)) % d.world) == d.rank; if (acc & !mine) atomicAdd(&red[WS_MISROUTED], 1ull); acc &= mine; }
        const u32 status = va.w & 0xFFFFu, proto = (va.w >> 16) & 0xFFu, dur = vb.x;
        const u32 err = is_error(proto, status);
        const u64 wt = (u64)vb.z | ((u64)vb.w << 32);
        st_acc += acc ? 1u : 0u;
        st_tmin = (acc && wt < st_tmin) ? wt : st_tmin; st_tmax = (acc && wt > st_tmax) ? wt : st_tmax;
        u32 Lm, Rm;
        sg_kmix(cf & nbmask, ct & nbmask, nbmask, &Lm, &Rm);        /* (the masks only matter for events that are not accepted) */
        const u32 part = Lm >> pshift, bucket = Rm & bmask;
        const u64 mk = ((u64)Lm << nb) | Rm;
        const ulonglong2 kk = reinterpret_cast<const ulonglong2*>(ckey)[bucket];
        spr = K1T_NONE; slo = 0; shi = 0;
        if (acc && !cache_fold(bucket, mk, kk.x, kk.y, (u64)dur, err)) {
            const u32 rank = atomicAdd(&bc[part], 1u);
            slo = dur; shi = ((u32)mk & rbmask) | (err << 31); spr = part | (rank << K1T_RANK_SHIFT);
        }
    };
    // WAITCNT: loads issued AFTER the group's eight (the prefetch of the next chunk) that may stay in flight
#define K1T_FOLD(WAITCNT, base, cend, bc, lo0, hi0, pr0, lo1, hi1, pr1, lo2, hi2, pr2, lo3, hi3, pr3)                 \
        { asm volatile("s_waitcnt vmcnt(" #WAITCNT ")" : "+v"(ea0), "+v"(eb0), "+v"(ea1), "+v"(eb1), "+v"(ea2), "+v"(eb2), "+v"(ea3), "+v"(eb3) : : "memory"); \
          bool r0, r1, r2, r3;                                                                                      \
          fast((base), (cend), ea0, eb0, (bc), r0, lo0, hi0, pr0); fast((base) + K1T_THREADS, (cend), ea1, eb1, (bc), r1, lo1, hi1, pr1); \
          fast((base) + 2 * K1T_THREADS, (cend), ea2, eb2, (bc), r2, lo2, hi2, pr2); fast((base) + 3 * K1T_THREADS, (cend), ea3, eb3, (bc), r3, lo3, hi3, pr3); \
          if (__builtin_amdgcn_ballot_w64(r0 | r1 | r2 | r3)) {             /* one copy of the general path: register selects */ \
              _Pragma("unroll 1")                                                                                   \
              for (u32 q = 0; q < 4; q++) {                                                                         \
                  const bool rq = q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : r3;                                     \
                  if (!__builtin_amdgcn_ballot_w64(rq)) continue;                                                   \
                  const v4u_t va = q == 0 ? ea0 : q == 1 ? ea1 : q == 2 ? ea2 : ea3;                                \
                  const v4u_t vb = q == 0 ? eb0 : q == 1 ? eb1 : q == 2 ? eb2 : eb3;                                \
                  u32 glo = 0, ghi = 0, gpr = K1T_NONE;                                                             \
                  if (rq) general(va, vb, (bc), glo, ghi, gpr);                                                     \
                  if (rq && q == 0) { lo0 = glo; hi0 = ghi; pr0 = gpr; }                                            \
                  if (rq && q == 1) { lo1 = glo; hi1 = ghi; pr1 = gpr; }                                            \
                  if (rq && q == 2) { lo2 = glo; hi2 = ghi; pr2 = gpr; }                                            \
                  if (rq && q == 3) { lo3 = glo; hi3 = ghi; pr3 = gpr; }                                            \
              } } }
    {
        // piece counters: zero by definition in the first batch of a window (no loads); a later batch reads them with
        // ordinary loads BEFORE anything is issued by hand
        for (u32 p = t; p < NP; p += K1T_THREADS) {
            uint2 h = make_uint2(0u, 0u);
            if (!first) h = d.hdr8[(size_t)p * d.nwg + w];
            fcn2[p] = h.x; fcn2[NP + p] = h.x; fcw[p] = h.y; bcnt[p] = 0u; bcnt[NP + p] = 0u;
        }
        for (u32 k = t; k < CT; k += K1T_THREADS) ckey[k] = SG_EKEY_EMPTY;
        for (u32 k = t; k < CT * 4; k += K1T_THREADS) cacc[k] = 0;
        if (t < 8) red[t] = t == WS_TMIN ? ~0ull : 0ull;
        // the join blob: six 16-byte loads per lane at most, issued and waited for in ONE asm statement (see k1a_partition: no
        // code may sit between a hand-issued load and the wait that names its registers)
        v4u_t jb0, jb1, jb2, jb3, jb4, jb5;
        static_assert(K1A_NJ == 6, "written out for 6 blob words per lane");
        const u32 n16 = d.jstage_bytes >> 4, n1 = (d.jl1mask + 1) >> 1;   // 16-byte words to stage; of them level 1 (always there)
        const uint4* g1 = reinterpret_cast<const uint4*>(d.jl1); const uint4* g2 = reinterpret_cast<const uint4*>(d.jl2) - n1;
#define K1T_JIDX(k) ((t + (k) * K1T_THREADS) < n16 ? (t + (k) * K1T_THREADS) : n16 - 1)
#define K1T_JSRC(k) ((K1T_JIDX(k) < n1 ? g1 : g2) + K1T_JIDX(k))
        const uint4* js0 = K1T_JSRC(0); const uint4* js1 = K1T_JSRC(1); const uint4* js2 = K1T_JSRC(2);
        const uint4* js3 = K1T_JSRC(3); const uint4* js4 = K1T_JSRC(4); const uint4* js5 = K1T_JSRC(5);
        asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %7, off\n\tglobal_load_dwordx4 %2, %8, off\n\t"
                     "global_load_dwordx4 %3, %9, off\n\tglobal_load_dwordx4 %4, %10, off\n\tglobal_load_dwordx4 %5, %11, off\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(jb0), "=&v"(jb1), "=&v"(jb2), "=&v"(jb3), "=&v"(jb4), "=&v"(jb5)
                     : "v"(js0), "v"(js1), "v"(js2), "v"(js3), "v"(js4), "v"(js5) : "memory");
        // (level-1 words go in as they are; a level-2 word = four u32 entries kind << 30 | id -> four u16 entries kind << 14 | id)
#define K1T_P16(x) ((((x) >> 30) << 14) | ((x) & 0x3FFFu))
#define K1T_JST(k, r) { const u32 i_ = t + (k) * K1T_THREADS;                                                          \
            if (i_ < n16) { if (L2M == 2 && i_ >= n1) reinterpret_cast<uint2*>(jl + n1)[i_ - n1] = make_uint2(K1T_P16((r).x) | (K1T_P16((r).y) << 16), K1T_P16((r).z) | (K1T_P16((r).w) << 16)); \
                            else jl[i_] = make_uint4((r).x, (r).y, (r).z, (r).w); } }
        K1T_JST(0, jb0); K1T_JST(1, jb1); K1T_JST(2, jb2); K1T_JST(3, jb3); K1T_JST(4, jb4); K1T_JST(5, jb5);
#undef K1T_JST
#undef K1T_P16
#undef K1T_JSRC
#undef K1T_JIDX
        LDS_BARRIER();
        SG_STAMP(d, 0, 1);
    }
    // PACKB: the partition number rides in the free bits [rb, rb + pb) of a parked record's high word (2 nb <= 31): the copy-out
    // is then one thread per tile position, all of a thread's LDS reads in flight together; otherwise 16 lanes walk a run.
    const bool packb = 2u * nb <= 31u;
    const u32 rb = d.rb;
    // P4 of a tile: copy its runs to the pieces — adjacent lanes, adjacent addresses.  It runs at the TOP of the next tile, behind
    // that tile's first eight loads: the loads' HBM latency (the barriers keep the waves in step, so nothing else would hide it)
    // passes while the previous tile's records leave.  Legal anywhere between the barrier behind P3 and the next barrier behind P1:
    // the tile, the offsets, this tile's run counters and piece counters are not written before that barrier.
    auto copy_out = [&](const u32 pc) {
        u32* bc = bcnt + pc * NP;
        u32* fcn = fcn2 + pc * NP;
    if (packb) {
        const u32 total = boff[NP - 1] + bc[NP - 1];             // records in the tile
        const u32 strip = ~(((1u << d.pb) - 1u) << rb);         // clears the partition bits (bit 31 = error stays)
#pragma unroll
        for (u32 k = 0; k < 4 * NSUB; k++) {
            const u32 i = t + k * K1T_THREADS;
            if (i < total) {
                const u64 rec = tile[i];
                const u32 hi = (u32)(rec >> 32), b = (hi >> rb) & ((1u << d.pb) - 1u);
                const u32 pos = fcn[b] + (i - boff[b]);
                const u64 out = (rec & 0xFFFFFFFFull) | ((u64)(hi & strip) << 32);
                if (pos < d.sn) { if (!SG_ABL(d, 0x1u)) piece8(d, b, w)[pos] = out; }
                else { K1T_LNEW(L); ovf8_single(d, b, ((u64)b << rb) | (hi & rbmask), rec & 0xFFFFFFFFull, hi >> 31, 0u, L); lflush(L); }
            }
        }
    } else {
        const u32 bpw = NP >> 4;                                 // partitions whose runs a wave writes out (np >= 64): four per step, 16 lanes each
        for (u32 b4 = 0; b4 < bpw; b4 += 4) {
            const u32 b = wave * bpw + b4 + (lane >> 4), j0 = lane & 15u;
            const u32 cnt = bc[b], off = boff[b], pos0 = fcn[b];
            u64* dst = piece8(d, b, w);
            for (u32 j = j0; j < cnt; j += 16) {
                const u64 rec = tile[off + j];
                const u32 pos = pos0 + j;
                if (pos < d.sn) { if (!SG_ABL(d, 0x1u)) dst[pos] = rec; }
                else { K1T_LNEW(L); ovf8_single(d, b, ((u64)b << rb) | ((u32)(rec >> 32) & rbmask), rec & 0xFFFFFFFFull, (u32)(rec >> 63), 0u, L); lflush(L); }
            }
        }
    }
    };
    bool havep = false; u32 pcur = 0;
    u32 cur = 0;
    u64 tk_p1 = 0, tk_wait = 0, tk_scan = 0, tk_p3 = 0, tk_b3 = 0, tk_p4 = 0;   // SG_ABLATE & 0x100: wave 0's clock ticks per phase
    for (u64 c0 = wr; c0 < nchunk; c0 += (u64)NSUB * d.nwg, cur ^= 1u) {
        u32* bc = bcnt + cur * NP;
        u32* fcn = fcn2 + cur * NP;                                  // counts before this tile; the scan writes fcn2[cur ^ 1] = counts behind it
        u32 lo0, hi0, pr0, lo1, hi1, pr1, lo2, hi2, pr2, lo3, hi3, pr3, lo4, hi4, pr4, lo5, hi5, pr5, lo6, hi6, pr6, lo7, hi7, pr7;
        const u64 tk0 = SG_ABL(d, 0x100u) ? wall_clock64() : 0ull;
        {   // P1: NSUB chunks of up to four events per thread
            v4u_t ea0, eb0, ea1, eb1, ea2, eb2, ea3, eb3;
            const u64 cb0 = c0 * chunk, ce0 = cb0 + chunk < end ? cb0 + chunk : end;
            const u64 i0 = cb0 + t;
            K1T_ISSUE(i0, ce0, cb0);
            if (havep) { const u64 tq = SG_ABL(d, 0x100u) ? wall_clock64() : 0ull; copy_out(pcur); if SG_ABL(d, 0x100u) tk_p4 += wall_clock64() - tq; }
            K1T_FOLD(0, i0, ce0, bc, lo0, hi0, pr0, lo1, hi1, pr1, lo2, hi2, pr2, lo3, hi3, pr3);
            if constexpr (NSUB == 2) {
                const u64 c1 = c0 + d.nwg;
                const u64 cb1 = c1 < nchunk ? c1 * chunk : cb0, ce1 = c1 < nchunk ? (cb1 + chunk < end ? cb1 + chunk : end) : cb0;   // no second chunk: an empty range
                const u64 i1 = cb1 + t;
                K1T_ISSUE(i1, ce1, cb1);
                K1T_FOLD(0, i1, ce1, bc, lo4, hi4, pr4, lo5, hi5, pr5, lo6, hi6, pr6, lo7, hi7, pr7);
            } else { pr4 = pr5 = pr6 = pr7 = K1T_NONE; lo4 = hi4 = lo5 = hi5 = lo6 = hi6 = lo7 = hi7 = 0; }
        }
        const u64 tk1 = SG_ABL(d, 0x100u) ? wall_clock64() : 0ull;
        LDS_BARRIER();
        const u64 tk2 = SG_ABL(d, 0x100u) ? wall_clock64() : 0ull;
        tk_p1 += tk1 - tk0; tk_wait += tk2 - tk1;
        {   // P2: exclusive scan of the run lengths by EVERY wave (sixteen identical scans cost less than a barrier behind one: the
            // offsets a wave needs in P3 are its own writes, every wave writes the same values).  Lane l owns the np / 64 consecutive
            // partitions from l * np / 64; the wave-wide part is DPP + readlane, no LDS round trip.
            // (16-byte LDS accesses: a lane's partitions are consecutive words — read one by one, 64 lanes 32 bytes apart hit four banks)
            const u32 pl = NP >> 6, b0 = lane * pl;
            const bool own = (b0 / (NP >> 4)) == wave;               // the wave that owns a partition (wave = partition / (np / 16)) also moves its piece
            u32* fnext = fcn2 + (cur ^ 1u) * NP;                     // counter on and re-arms the run counters tile k + 1's ranks come from (tile k - 1's
            u32* bprev = bcnt + (cur ^ 1u) * NP;                     // set: read for the last time in its scan, every wave is past that)
            if ((pl & 3u) == 0 && pl <= 16) {
                const u32 nq = pl >> 2;                              // 1, 2 or 4 quads per lane (np = 256, 512, 1024)
                uint4 cq[4];
                u32 s = 0;
#pragma unroll
                for (u32 k = 0; k < 4; k++) { cq[k] = k < nq ? reinterpret_cast<const uint4*>(bc + b0)[k] : make_uint4(0u, 0u, 0u, 0u); s += cq[k].x + cq[k].y + cq[k].z + cq[k].w; }
                u32 incl = s;                                        // inclusive scan over the 64 lanes: DPP row_shr 1, 2, 4, 8 (zero fill), then the row totals
                incl += dpp32<0x111>(incl); incl += dpp32<0x112>(incl); incl += dpp32<0x114>(incl); incl += dpp32<0x118>(incl);
                const u32 r0 = rdlane32(incl, 15), r1 = rdlane32(incl, 31), r2 = rdlane32(incl, 47);
                incl += (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
                u32 run = incl - s;
#pragma unroll
                for (u32 k = 0; k < 4; k++) if (k < nq) {
                    uint4 o; o.x = run; run += cq[k].x; o.y = run; run += cq[k].y; o.z = run; run += cq[k].z; o.w = run; run += cq[k].w;
                    reinterpret_cast<uint4*>(boff + b0)[k] = o;
                    if (own) {
                        const uint4 f = reinterpret_cast<const uint4*>(fcn + b0)[k];
                        uint4 nx; nx.x = f.x + cq[k].x; nx.y = f.y + cq[k].y; nx.z = f.z + cq[k].z; nx.w = f.w + cq[k].w;
                        nx.x = nx.x < d.sn ? nx.x : d.sn; nx.y = nx.y < d.sn ? nx.y : d.sn; nx.z = nx.z < d.sn ? nx.z : d.sn; nx.w = nx.w < d.sn ? nx.w : d.sn;
                        reinterpret_cast<uint4*>(fnext + b0)[k] = nx;
                        reinterpret_cast<uint4*>(bprev + b0)[k] = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            } else {
                u32 s = 0;
                for (u32 k = 0; k < pl; k++) s += bc[b0 + k];
                u32 incl = s;
                incl += dpp32<0x111>(incl); incl += dpp32<0x112>(incl); incl += dpp32<0x114>(incl); incl += dpp32<0x118>(incl);
                const u32 r0 = rdlane32(incl, 15), r1 = rdlane32(incl, 31), r2 = rdlane32(incl, 47);
                incl += (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
                u32 run = incl - s;
                for (u32 k = 0; k < pl; k++) {
                    const u32 c = bc[b0 + k]; boff[b0 + k] = run; run += c;
                    if (own) { const u32 nx = fcn[b0 + k] + c; fnext[b0 + k] = nx < d.sn ? nx : d.sn; bprev[b0 + k] = 0u; }
                }
            }
        }
        const u64 tk3 = SG_ABL(d, 0x100u) ? wall_clock64() : 0ull;
        // P3: every thread drops its records at offset + rank (PACKB: with the partition number in the free bits of the high word)
#define K1T_DROP(lo, hi, pr) if ((pr) != K1T_NONE) { const u32 pt_ = (pr) & ((1u << K1T_RANK_SHIFT) - 1u);                                  \
            tile[boff[pt_] + ((pr) >> K1T_RANK_SHIFT)] = (u64)(lo) | ((u64)((hi) | (packb ? pt_ << rb : 0u)) << 32); }
        K1T_DROP(lo0, hi0, pr0); K1T_DROP(lo1, hi1, pr1); K1T_DROP(lo2, hi2, pr2); K1T_DROP(lo3, hi3, pr3);
        if constexpr (NSUB == 2) { K1T_DROP(lo4, hi4, pr4); K1T_DROP(lo5, hi5, pr5); K1T_DROP(lo6, hi6, pr6); K1T_DROP(lo7, hi7, pr7); }
#undef K1T_DROP
        const u64 tk4 = SG_ABL(d, 0x100u) ? wall_clock64() : 0ull;
        LDS_BARRIER();
        const u64 tk5 = SG_ABL(d, 0x100u) ? wall_clock64() : 0ull;
        havep = true; pcur = cur;
        if SG_ABL(d, 0x100u) { tk_scan += tk3 - tk2; tk_p3 += tk4 - tk3; tk_b3 += tk5 - tk4; }
    }
    if (havep) copy_out(pcur);                                       // the last tile's runs
    u32* fcn = fcn2 + cur * NP;                                      // the counts behind the last tile (written by its scan)
    SG_STAMP(d, 0, 3);
    if (SG_ABL(d, 0x100u) && t == 0 && blockIdx.x < 4096) { u64* g = d.dbg + ((size_t)2 * 4096 + blockIdx.x) * 8; g[0] = tk_p1; g[1] = tk_wait; g[2] = tk_scan; g[3] = tk_p3; g[4] = tk_b3; g[5] = tk_p4; }
#undef K1T_ISSUE
#undef K1T_FOLD
    LDS_BARRIER();
    SG_STAMP(d, 0, 4);
    // flush the cache: a key seen once leaves as a single record, the others as aggregates
    K1T_LNEW(L);
    for (u32 s = t; s < CT; s += K1T_THREADS) {
        const u64 k = ckey[s];
        if (k == SG_EKEY_EMPTY) continue;
        const u64 x0 = cacc[s * 4];
        const u32 part = (u32)(k >> d.rb), rem = (u32)k & rbmask;
        if ((x0 & 0xFFFFFFFFull) == 1ull) {
            const u64 dur = cacc[s * 4 + 1];
            if (dur >> 32) emit_wide(d, fcw, w, part, k, dur, (u32)(x0 >> 32), 0u, L);
            else emit_narrow_direct(d, fcn, w, part, rem, (u32)dur, (u32)(x0 >> 32), L);
        } else if ((x0 & 0xFFFFFFFFull) != 0ull) emit_agg8(d, fcw, w, part, k, x0, cacc[s * 4 + 1], cacc[s * 4 + 2], cacc[s * 4 + 3], L, first);
    }
    // workgroup statistics: wave reduce -> LDS -> one thread updates this workgroup's private line
    lflush(L);
    {
        const u64 tmin = wave_min_u64(st_tmin), tmax = wave_max_u64(st_tmax);
        const u32 ac = wave_sum_u32(st_acc);
        if (lane == 0 && ac) { atomicMin(&red[WS_TMIN], tmin); atomicMax(&red[WS_TMAX], tmax); atomicAdd(&red[WS_ACCEPTED], (u64)ac); }
    }
    LDS_BARRIER();
    for (u32 p = t; p < NP; p += K1T_THREADS) d.hdr8[(size_t)p * d.nwg + w] = make_uint2(fcn[p], fcw[p]);
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
    if (clk_me) { atomicAdd(&d.clk[0], __builtin_readcyclecounter() - clk_c0); atomicAdd(&d.clk[1], wall_clock64() - clk_r0); }
#undef LDS_BARRIER
#undef K1T_LNEW
}

// ---- pass B ---------------------------------------------------------------------------------------------------------
// Workgroup q = p * S + s owns SUB-TABLE s of partition p (S = d.k1b_split = 1 or 2; the top remainder bit picks the
// sub-table): pass A wants few partitions (long runs per tile), pass B wants tables small enough for two workgroups per CU
// — with S = 2 both workgroups of a partition read all of its records (the second read comes out of L2 / the Infinity Cache:
// the two are 8 blocks apart, i.e. on the same XCD, and start together) and each merges the half that is its own.
// A workgroup reads the record counts of the partition's nwg pieces, then exactly the records that exist — U 16-byte loads
// (two narrow records each) per lane in flight, the lanes of a piece side by side —, merges them in an LDS table keyed by the
// u32 remainder and writes every distinct edge once with plain stores; the endpoints come back out of the key with
// sg_kunmix.  Outputs as k1b_merge, per OUTPUT partition q: e_from / e_to / acc_src / e_rank, part_n[q], deg[from][replica].
// Narrow records first — their durations are below 2^32, so count, error count and max are 32-bit LDS atomics on the low
// words of the accumulators —, then, behind a barrier, the wide singles, aggregates and overflow records with 64-bit ones.
// LDS: hacc[4][HT] u64 | hkey[HT] u32 | two counters: 36 bytes per slot.  SPT = table slots per thread in the compaction
// (HT / threads): one round, all returning `deg` atomics of a thread in flight together.
// PACK (round 4): a narrow record's count and duration travel in ONE 64-bit LDS add — count in bits 48.., the duration sum below —
// instead of a 32-bit add and a 64-bit add.  The merge runs at the rate of the LDS atomic unit (§3 K1: ~1.5 lane-operations per clock and
// CU, 5.45 per record), so an operation less per record is time.  Exact as long as a workgroup merges fewer than 2^16 narrow records —
// every one is below 2^32 ns, so the sum stays below 2^48 and cannot carry into the count —, which the host guarantees from the geometry
// (sn x nwg < 65 536: a piece holds at most sn narrow records) before it picks this build; behind the narrow phase's barrier every slot's
// owner thread moves the count into accumulator 0, where the wide records (64-bit adds) expect it.
// WM (warm windows, sg_device.h): 0 = an engine without the kept state: the kernel as it was.  1 = the WARM attempt: the table starts as
// the image the last cold window left (wk_keys), a record finds its key at the slot it had then, and at the end every seeded slot's
// accumulators go straight to k_acc[wk_pos[slot]] — zero for a slot no record touched, bit 63 of the max word set for one that was
// (an SG_EV_ALIVE record touches a key without counting: its own LDS bit map says so).  A key the image lacks is inserted like any
// other, which makes the window COLD (C_COLD): the full rebuild repeats the merge.  A window already known to be cold is skipped.
// 2 = the COLD merge of an engine that keeps state: skipped on a warm window.  As WM 0 for the window's records; then, behind them, the
// keys of the OLD image that no record touched are put into the table too as long as the partition's output has room (they can
// never displace a key of this window: those are all in by then) — the kept set is the UNION of what the windows have touched, so a
// stream whose windows each touch another 90 % of the graph settles on the warm path instead of rebuilding for ever.  Every key
// leaves with its accumulators (bit 63 of the max word = touched in this window, all zero otherwise); the image and, per slot, the
// edge's index in the partition output are left for kw_capture.  The rebuild (K2) then builds the KEPT CSR from these outputs and
// kw_compact derives the window's CSR from it, exactly as on a warm window.
template <int U, int SPT, bool PACK, int WM>
__device__ __forceinline__ void k1b8_body(const Dev& d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (WM == 1) { if (d.ctr[C_COLD]) return; }             // (uniform: kc_prepare already knows, or another workgroup found a new key)
    if constexpr (WM == 2) { if (!d.ctr[C_COLD]) return; }
    const u32 HT = d.k1b_ht, hmask = HT - 1;
    u64* hacc = reinterpret_cast<u64*>(smem);                        // [4][HT]: accumulator j of slot h at j*HT + h
    u32* hkey = reinterpret_cast<u32*>(hacc + (size_t)HT * 4);       // [HT] remainders, all ones = empty
    u32* n_drop = hkey + HT; u32* out_n = n_drop + 1;
    u32* htouch = out_n + 1;                                         // WM 1: [HT / 32] slots a record touched without counting; word HT / 32 = new keys inserted
    u32* hnew = htouch + HT / 32 + 1;                                // WM 1: [HT / 32] slots whose key the image lacked (inserted by this window's records)
    u32* nkeys = hnew + HT / 32;                                     // WM 1: keys in the table (the image's + the new ones): a partition keeps at most pcap
    const u32 t = threadIdx.x, NT = blockDim.x;
    // q -> (partition, sub-table): blocks b and b + 8 run on the same XCD (b % 8) and are dispatched back to back
    const u32 S = d.k1b_split, q = blockIdx.x;
    const u32 pi = S == 2 ? ((q >> 4) << 3) | (q & 7u) : q, sidx = S == 2 ? (q >> 3) & 1u : 0u;
    const u32 p = d.k1b_order ? d.k1b_order[pi] : pi;               // the partitions in the order of their size in the window before, largest first (kc_prepare)
    const u32 oq = p * S + sidx;                                     // output partition
    const u32 nb = d.nb, nbmask = (1u << nb) - 1u, rbmask = (1u << d.rb) - 1u, sbit = d.rb - 1;
    SG_STAMP(d, 1, 0);
    if (SG_ABL(d, 0x100u) && threadIdx.x == 0 && blockIdx.x < 4096)   // where the workgroup runs: HW_ID (CU 8..11, SH 12, SE 13..15) | XCC_ID << 32
        d.dbg[((size_t)1 * 4096 + blockIdx.x) * 8 + 7] = (u64)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((u64)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);
    const bool empty = d.batch_state == 2u;                          // no batch this window: the pieces are the previous window's
    const u32 LPP = NT > d.nwg ? NT / d.nwg : 1u;                    // lanes per piece
    const u32 sub = t % LPP, w0 = t / LPP;
    uint2 h0 = make_uint2(0u, 0u);
    u32 have_keys = 0;                                               // WM 1: keys of the image this thread seeded
    if (!empty && w0 < d.nwg) h0 = d.hdr8[(size_t)p * d.nwg + w0];
    // the first U record pairs of the lane's first piece go out together with the header (index clamped to the narrow region;
    // what lies beyond the count is ignored): one round trip instead of two, hidden behind the table set-up
    uint4 xf[U];
    {
        const uint4* piece0 = reinterpret_cast<const uint4*>(piece8(d, p, w0 < d.nwg ? w0 : 0u));
        const u32 pm1 = (d.sn >> 1) - 1;
#pragma unroll
        for (int u = 0; u < U; u++) { const u32 r = sub + (u32)u * LPP; xf[u] = piece0[r < pm1 ? r : pm1]; }
    }
    const u64 ovf_n = d.ctr[C_OVF_N];
    const u32 nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS], nob = (u32)d.ctr[C_N_OBIP];
    if constexpr (WM == 1) {
        const u32* img = d.wk_keys + (size_t)oq * HT;                 // (coalesced: 4 HT bytes per workgroup, beside the headers' round trip)
        if (t == 0) *nkeys = 0;
        for (u32 i = t; i <= HT / 32; i += NT) htouch[i] = 0;
        for (u32 i = t; i < HT / 32; i += NT) hnew[i] = 0;
        for (u32 i = t; i < HT; i += NT) { const u32 k = img[i]; hkey[i] = k; have_keys += k != 0xFFFFFFFFu ? 1u : 0u; }
    } else {
        for (u32 i = t; i < HT; i += NT) hkey[i] = 0xFFFFFFFFu;
        if constexpr (WM == 2) { for (u32 i = t; i <= HT / 32; i += NT) htouch[i] = 0; }   // bits: slots filled from the old image; word HT / 32: keys in the table
    }
    for (u32 i = t; i < HT * 4; i += NT) hacc[i] = 0;
    if (t == 0) { *n_drop = 0; *out_n = 0; }
    __syncthreads();
    if constexpr (WM == 1) { have_keys = wave_sum_u32(have_keys); if ((t & 63u) == 0 && have_keys) atomicAdd(nkeys, have_keys); }   // (read at the end, behind barriers)
    SG_STAMP(d, 1, 1);

    // slot of a key (find or insert), or HT when the table is full
    auto slot_of = [&](u32 rem) -> u32 {
        u32 h = rem & hmask;
        for (u32 it = 0; it < HT; it++) {
            u32 k = lds_fresh_u32(&hkey[h]);
            if (k == 0xFFFFFFFFu) {
                k = atomicCAS(&hkey[h], 0xFFFFFFFFu, rem);
                if (k == 0xFFFFFFFFu) {
                    k = rem;
                    if constexpr (WM == 1) { atomicOr(&hnew[h >> 5], 1u << (h & 31u)); atomicAdd(&htouch[HT / 32], 1u); }   // a key the kept set lacks: a NEW edge of this window (delta, below)
                    if constexpr (WM == 2) atomicAdd(&htouch[HT / 32], 1u);   // keys in the table (the union phase fills what room is left)
                }
            }
            if (k == rem) return h;
            h = (h + 1) & hmask;
        }
        return HT;
    };
    auto mine = [&](u32 rem) -> bool { return S == 1 || ((rem >> sbit) & 1u) == sidx; };
    u32* hacc32 = reinterpret_cast<u32*>(hacc);                       // low word of accumulator j of slot h: 2 * (j*HT + h)
    // (measured on one box each: both sums as 32-bit words with a carry, 125.6-130.0 vs 121.3-129.3 us; reading the maximum first and
    //  sending the atomic only when it would rise, 128.0-129.2 vs 122.6-123.5 us: the merge is not bound by the number or width of
    //  its LDS atomics.  Without the returning `deg` atomics of the compaction the launch is 91 us instead of 114; without any merge 48.)
    // the two records of one 16-byte load together: their first probes are in flight at once (a hit at the home slot — seven in ten at this
    // load — then costs the pair one LDS round trip instead of two); whatever the first probe does not settle walks the probe loop as before
    auto apply_narrow = [&](u32 lo, u32 hi, u32 h) {
        if (h == HT) { atomicAdd(n_drop, 1u); return; }
        const u32 us = div1000_u32(lo);
        if constexpr (PACK) atomicAdd(&hacc[HT + h], (u64)lo | (1ull << 48));
        else { atomicAdd(&hacc32[2 * h], 1u); atomicAdd(&hacc[HT + h], (u64)lo); }
        if (hi >> 31) atomicAdd(&hacc32[2 * h + 1], 1u);
        atomicMax(&hacc32[2 * (2 * HT + h)], lo);
        atomicAdd(&hacc[3 * HT + h], (u64)us * (u64)us);
    };
    // a slot that read EMPTY: take it (or learn whose it has become); returns the key the slot holds now
    auto claim = [&](u32 h, u32 rem) -> u32 {
        u32 k = atomicCAS(&hkey[h], 0xFFFFFFFFu, rem);
        if (k == 0xFFFFFFFFu) {
            k = rem;
            if constexpr (WM == 1) { atomicOr(&hnew[h >> 5], 1u << (h & 31u)); atomicAdd(&htouch[HT / 32], 1u); }
            if constexpr (WM == 2) atomicAdd(&htouch[HT / 32], 1u);
        }
        return k;
    };
#ifdef SG_K1B_PROBE_R5                                            /* (round 5's form, for a same-box A/B of two builds) */
    auto add_narrow2 = [&](u32 lo0, u32 hi0, u32 lo1, u32 hi1, bool second) {
        const u32 rem0 = hi0 & rbmask, rem1 = hi1 & rbmask;
        const bool a0 = mine(rem0) && !SG_ABL(d, 0x10u), a1 = second && mine(rem1) && !SG_ABL(d, 0x10u);
        const u32 h0 = rem0 & hmask, h1 = rem1 & hmask;
        const u32 k0 = lds_fresh_u32(&hkey[h0]), k1 = lds_fresh_u32(&hkey[h1]);
        if (a0) apply_narrow(lo0, hi0, k0 == rem0 ? h0 : slot_of(rem0));
        if (a1) apply_narrow(lo1, hi1, k1 == rem1 ? h1 : slot_of(rem1));
    };
#else
    // Round 6 (late): the two records walk their probe chains in ONE loop whose branches are wave-uniform (a ballot decides; a lane that has
    // arrived re-reads its slot), so the lanes' state lives in scalar masks and the exec mask is not rebuilt every trip.  Measured against
    // round 5's per-record loop on one box (profiles/r06_probe_ab_c3.txt, v0..v4): the launch takes the same 97-99 us either way — 25.3 M
    // vector wave-instructions per launch in both forms, 15.6 M scalar ones instead of 18.7 M, twice the LDS bank-conflict cycles
    // (profiles/r06_pair_k1_sq_counters_c3.txt against r06_final_*).  Kept for its registers (48 with the rolling buffer below, 59 before).
    // What the experiment settled: the merge is bound neither by its probe chains nor by the LDS atomic unit (tools/lds_atomic_probe: a
    // record's four operations run at 1.5 records per clock and CU on a table like this one, four times what pass B needs) nor by its loads
    // (the rolling buffer changed nothing), and one table per partition (SG_SPLIT=1: every lane busy, half the waves) takes the same time too;
    // the vector unit is three quarters busy while two workgroups share a CU, and the launch is two rounds of such workgroups plus a tail
    // in which the CUs' last workgroups run alone (tools/stamps.py, profiles/r06_blockorder_stamps_c3.txt).
    auto add_narrow2 = [&](u32 lo0, u32 hi0, u32 lo1, u32 hi1, bool second) {
        const u32 rem0 = hi0 & rbmask, rem1 = hi1 & rbmask;
        const bool a0 = mine(rem0) && !SG_ABL(d, 0x10u), a1 = second && mine(rem1) && !SG_ABL(d, 0x10u);
        u32 h0 = rem0 & hmask, h1 = rem1 & hmask;
        u32 k0 = lds_fresh_u32(&hkey[h0]), k1 = lds_fresh_u32(&hkey[h1]);
        bool s0 = a0, s1 = a1;                                       // still looking
        for (u32 left = HT;;) {
            const bool e0 = s0 && k0 == 0xFFFFFFFFu, e1 = s1 && k1 == 0xFFFFFFFFu;
            if (__builtin_amdgcn_ballot_w64(e0 || e1)) {             // (rare on a warm window: a key the table lacks)
                if (e0) k0 = claim(h0, rem0);
                if (e1) k1 = claim(h1, rem1);
            }
            s0 = s0 && k0 != rem0; s1 = s1 && k1 != rem1;
            if (!__builtin_amdgcn_ballot_w64(s0 || s1) || --left == 0) break;   // (left == 0: the table is full — what is still looking is dropped)
            h0 = s0 ? (h0 + 1) & hmask : h0; h1 = s1 ? (h1 + 1) & hmask : h1;
            k0 = lds_fresh_u32(&hkey[h0]); k1 = lds_fresh_u32(&hkey[h1]);   // (reads by the lanes still looking only, under exec masks: measured, no difference)
        }
        if (a0) apply_narrow(lo0, hi0, s0 ? HT : h0);
        if (a1) apply_narrow(lo1, hi1, s1 ? HT : h1);
    };
#endif
    auto add_wide = [&](u32 rem, u64 a0, u64 a1, u64 a2, u64 a3) {
        if (!mine(rem)) return;
        const u32 h = slot_of(rem);
        if (h == HT) { atomicAdd(n_drop, (u32)(a0 & 0xFFFFFFFFull)); return; }
        atomicAdd(&hacc[h], a0); atomicAdd(&hacc[HT + h], a1); atomicMax(&hacc[2 * HT + h], a2); atomicAdd(&hacc[3 * HT + h], a3);
        if constexpr (WM == 1) { if (!(a0 & 0xFFFFFFFFull)) atomicOr(&htouch[h >> 5], 1u << (h & 31u)); }   // an edge-only record: the edge exists in this window with count 0
    };
    u32 my_nn = 0;
    for (u32 w = w0; w < d.nwg; w += NT / LPP) {
        const uint2 h = empty ? make_uint2(0u, 0u) : (w == w0 ? h0 : d.hdr8[(size_t)p * d.nwg + w]);   // (no batch this window: the headers are the previous window's)
        const u32 nn = h.x < d.sn ? h.x : d.sn;
        if (!nn) continue;
        const uint4* pairs = reinterpret_cast<const uint4*>(piece8(d, p, w));
        const u32 npair = (nn + 1) >> 1, lastp = npair - 1;
        if (SG_ABL(d, 0x100u) && sub == 0 && blockIdx.x < 4096) atomicAdd(&d.dbg[((size_t)1 * 4096 + blockIdx.x) * 8 + 6], (u64)nn);   // (tools/stamps.py: narrow records of the partition)
        if (sub == 0) my_nn += nn;
#ifdef SG_K1B_NO_PREFETCH                                                /* (round 5's form: a round's loads are waited for before its merges begin) */
        for (u32 r0 = sub; r0 < npair; r0 += LPP * U) {
            uint4 x[U];
            if (w == w0 && r0 == sub) {
#pragma unroll
                for (int u = 0; u < U; u++) x[u] = xf[u];
            } else {
#pragma unroll
                for (int u = 0; u < U; u++) { const u32 r = r0 + u * LPP; x[u] = pairs[r < npair ? r : lastp]; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const u32 r = r0 + u * LPP;
                if (r < npair) add_narrow2(x[u].x, x[u].y, x[u].z, x[u].w, 2 * r + 1 < nn);
            }
        }
#else
        // a rolling buffer of U pairs: a slot is refilled with the pair the lane needs a round later as soon as its content has been taken —
        // the load's round trip passes behind the merges of the other slots instead of in front of the round.  (Measured: no change in the
        // launch's time — other waves already covered the wait —; kept because U = 2 in this form needs 48 registers where U = 4 needed 59.)
        uint4 x[U];
        if (w == w0) {
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = xf[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) { const u32 r = sub + u * LPP; x[u] = pairs[r < npair ? r : lastp]; }
        }
        for (u32 r0 = sub; r0 < npair; r0 += LPP * U) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const u32 r = r0 + u * LPP, rn = r + LPP * U;
                const uint4 c = x[u];
                x[u] = pairs[rn < npair ? rn : lastp];
                if (r < npair) add_narrow2(c.x, c.y, c.z, c.w, 2 * r + 1 < nn);
            }
        }
#endif
    }
    if (sidx == 0 && d.k1b_cnt) {                                    // the partition's records, for the next window's order (one add per wave)
        my_nn = wave_sum_u32(my_nn);
        if ((t & 63u) == 0 && my_nn) atomicAdd(&d.k1b_cnt[p], my_nn);
    }
    __syncthreads();                                                 // every 32-bit max is in: 64-bit updates may follow
    SG_STAMP(d, 1, 3);
    // (WM 1: until round 5 the workgroup looked at C_COLD again here and left when another one had given up — each wave for itself, so
    // that waves which saw different values parted ways in front of the barriers below.  Since round 6 an unknown key no
    // longer makes a window cold; what still does — a full table, a dropped edge — is rare, and a workgroup that finishes a merge nobody
    // will read costs less than a dependent global load in front of every warm workgroup's barrier.)
    if constexpr (PACK) {                                            // unpack: a slot by one thread, nobody else touches the table here
        for (u32 i = t; i < HT; i += NT) { const u64 w = hacc[HT + i]; hacc[i] += w >> 48; hacc[HT + i] = w & ((1ull << 48) - 1ull); }
        __syncthreads();
    }
    for (u32 w = w0; w < d.nwg; w += NT / LPP) {
        const uint2 h = empty ? make_uint2(0u, 0u) : (w == w0 ? h0 : d.hdr8[(size_t)p * d.nwg + w]);
        const u32 nw = (h.y & 0xFFFFu) < d.sw ? (h.y & 0xFFFFu) : d.sw, na = (h.y >> 16) < d.sa ? (h.y >> 16) : d.sa;
        if (!(nw | na)) continue;
        const u64* piece = piece8(d, p, w);
        for (u32 r = sub; r < nw; r += LPP) {
            const uint4 x = reinterpret_cast<const uint4*>(piece + d.sn)[r];
            const u32 dhi = x.w & 0x3FFFFFFFu;
            const u64 dur = (u64)x.z | ((u64)dhi << 32);
            const u64 us = dur / 1000ull;
            const u64 one = ((x.w >> 30) & 1u) ? 0ull : 1ull;            // bit 62: edge-only record (SG_EV_ALIVE)
            add_wide(x.x & rbmask, one | ((u64)(x.w >> 31) << 32), dur, dur, us * us);
        }
        for (u32 r = sub; r < na; r += LPP) {
            const u64* a = piece + d.sn + 2 * d.sw + 5 * r;
            add_wide((u32)a[0] & rbmask, a[1], a[2], a[3], a[4]);
        }
    }
    if (ovf_n) {
        const u64 no = ovf_n < d.ovf_cap ? ovf_n : d.ovf_cap;
        for (u64 i = t; i < no; i += NT) {
            if (d.ovf_p[i] != p) continue;
            const u64* o = d.ovf + i * 9;
            add_wide((u32)o[0] & rbmask, o[1], o[2], o[3], o[4]);
        }
    }
    __syncthreads();
    SG_STAMP(d, 1, 4);
    if constexpr (WM == 2) {
        // the union: keys of the old image that this window did not touch, while the partition's output has room.  (A kept state that
        // is not whole — it was captured from a window with raw outbound IPs, whose compact indices are table slots of THAT window —
        // is not carried.)
        if (d.ctr[C_KEPT_VALID]) {                                   // (uniform)
            const u32* img = d.wk_keys + (size_t)oq * HT;
            u32 old[SPT];
#pragma unroll
            for (int k2 = 0; k2 < SPT; k2++) { const u32 sl = t + (u32)k2 * NT; old[k2] = sl < HT ? img[sl] : 0xFFFFFFFFu; }
#pragma unroll
            for (int k2 = 0; k2 < SPT; k2++) {
                const u32 rem = old[k2];
                if (rem == 0xFFFFFFFFu) continue;
                bool have = false, room = false;
                u32 h = rem & hmask;
                for (u32 it = 0; it < HT; it++) {
                    u32 k = lds_fresh_u32(&hkey[h]);
                    if (k == 0xFFFFFFFFu) {
                        if (!room) {                                 // absent so far: take one of the output's free places, or leave the key behind
                            if (atomicAdd(&htouch[HT / 32], 1u) >= d.pcap) { atomicSub(&htouch[HT / 32], 1u); break; }
                            room = true;
                        }
                        k = atomicCAS(&hkey[h], 0xFFFFFFFFu, rem);
                        if (k == 0xFFFFFFFFu) { atomicOr(&htouch[h >> 5], 1u << (h & 31u)); have = true; break; }   // in, marked "not of this window"
                    }
                    if (k == rem) { have = true; break; }            // (a record of this window has it)
                    h = (h + 1) & hmask;
                }
                if (room && !have) atomicSub(&htouch[HT / 32], 1u);  // (the table itself was full)
            }
        }
        __syncthreads();                                             // the old image has been read: the compaction below writes the new one
    }
    if constexpr (WM == 1) {
        // warm output: slot -> kept position (one coalesced read), 32 bytes to k_acc[position] — untouched slots write zeros, so the
        // kept array is rewritten whole every window and kw_compact needs no other mark
        const u32 KE = (u32)d.ctr[C_KEPT_E];
        const u32* wp = d.wk_pos + (size_t)oq * HT;
        u32 pos[SPT];
#pragma unroll
        for (int k2 = 0; k2 < SPT; k2++) { const u32 sl = t + (u32)k2 * NT; pos[k2] = sl < HT ? wp[sl] : SG_NONE; }
        bool cold = false;
        const bool anynew = htouch[HT / 32] != 0;                    // (uniform)
#pragma unroll
        for (int k2 = 0; k2 < SPT; k2++) {
            const u32 sl = t + (u32)k2 * NT;
            if (sl >= HT) continue;
            const u32 rem = hkey[sl];
            if (rem == 0xFFFFFFFFu) continue;
            const u64 a0 = hacc[sl], a1 = hacc[HT + sl], a2 = hacc[2 * HT + sl], a3 = hacc[3 * HT + sl];
            if (anynew && ((hnew[sl >> 5] >> (sl & 31u)) & 1u)) {
                // A NEW edge (round 6): it leaves in the cold format — endpoints out of the key, a place in the partition's output, its rank
                // among the new edges of its row (deg2) — and enters the image at once; kw_compact gives it its kept position.  A
                // partition that would keep more than pcap keys gives up (cold: the rebuild repeats the merge and does the accounting).
                const u64 mk = ((u64)p << d.rb) | rem;
                u32 cf, ct;
                sg_kunmix((u32)(mk >> nb), (u32)mk & nbmask, nbmask, &cf, &ct);
                const u32 f = cf, to = ct;                           // COMPACT ids (sg_kept_compact: a warm window has no raw outbound IP), as the kept CSR holds them
                if (atomicAdd(nkeys, 1u) >= d.pcap) { cold = true; continue; }
                const u32 di = atomicAdd(out_n, 1u);                 // (< pcap: the partition's keys are)
                const size_t slot = (size_t)oq * d.pcap + di;
                const u32 rk = atomicAdd(&d.deg2[SG_DEG_IDX(f, oq & (SG_DEG_REP - 1))], 1u);   // (ranking a row's new edges by a cursor in the delta scatter
                                                                     //  instead was measured: no gain here, + 40 us when the new edges come as whole new rows)
                d.e_from[slot] = f; d.e_to[slot] = to; d.e_rank[slot] = rk;
                ulonglong2* o = reinterpret_cast<ulonglong2*>(d.acc_src + slot * 4);
                o[0] = make_ulonglong2(a0, a1); o[1] = make_ulonglong2(a2 | (1ull << 63), a3);
                d.dl_img[slot] = oq * HT + sl;
                d.wk_keys[(size_t)oq * HT + sl] = rem;               // (its position: kw_compact, from dl_img)
                continue;
            }
            const bool touched = (a0 & 0xFFFFFFFFull) != 0 || ((htouch[sl >> 5] >> (sl & 31u)) & 1u);
            if (pos[k2] >= KE) { if (touched) cold = true; continue; }   // a key without a kept edge (dropped for capacity when the image was taken): only a window that
                                                                     // TOUCHES it must be rebuilt — flagged unconditionally it kept its partition cold for ever
            ulonglong2* o = reinterpret_cast<ulonglong2*>(d.k_acc + (size_t)pos[k2] * 4);
            o[0] = make_ulonglong2(touched ? a0 : 0ull, touched ? a1 : 0ull);
            o[1] = make_ulonglong2(touched ? (a2 | (1ull << 63)) : 0ull, touched ? a3 : 0ull);
        }
        if (cold || (t == 0 && *n_drop)) d.ctr[C_COLD] = 2;           // (same value from whoever writes it; 2 = found HERE, after a whole merge: the host backs off when that keeps happening)
        if (anynew) {                                                // (uniform)
            __syncthreads();                                         // the new edges are counted
            if (t == 0) { const u32 dn = *out_n; d.part_n[oq] = dn; if (dn) d.ctr[C_DELTA_N] = 1; }   // new edges of this partition (the delta chain reads it); C_DELTA_N: a plain
                                                                     // store of "there are some" — a thousand workgroups' atomic adds on one word queue up behind each other at the L2;
                                                                     // the delta chain's row scan stores the count
        } else if (t == 0) d.part_n[oq] = 0;
        SG_STAMP(d, 1, 5);
        return;
    }
    // compaction: SPT table slots per thread, one round; the returning `deg` atomics of a thread are in flight together
    {
        u32 f[SPT], to[SPT], oi[SPT], rk[SPT]; bool live[SPT];
#pragma unroll
        for (int k2 = 0; k2 < SPT; k2++) {
            const u32 sl = t + (u32)k2 * NT;
            live[k2] = false; f[k2] = to[k2] = oi[k2] = rk[k2] = 0;
            if (sl >= HT) continue;
            const u32 rem = hkey[sl];
            if constexpr (WM == 2) { d.wk_keys[(size_t)oq * HT + sl] = rem; d.wk_pos[(size_t)oq * HT + sl] = SG_NONE; }   // the image; the live slots' index follows below
            if (rem == 0xFFFFFFFFu) continue;
            const u64 mk = ((u64)p << d.rb) | rem;
            u32 cf, ct;
            sg_kunmix((u32)(mk >> nb), (u32)mk & nbmask, nbmask, &cf, &ct);
            if (WM == 2 && d.kept_compact && nob == 0) { f[k2] = cf; to[k2] = ct; }   // the KEPT state is built in compact ids (sg_kernels.h sg_kept_compact)
            else { f[k2] = dense_of(d, ref_of_ci(d, cf), nk, nl, nob); to[k2] = dense_of(d, ref_of_ci(d, ct), nk, nl, nob); }
            if (f[k2] == SG_NONE || to[k2] == SG_NONE) { atomicAdd(n_drop, (u32)(hacc[sl] & 0xFFFFFFFFull)); continue; }
            oi[k2] = atomicAdd(out_n, 1u);
            if (oi[k2] >= d.pcap) { atomicAdd(n_drop, (u32)(hacc[sl] & 0xFFFFFFFFull)); continue; }
            live[k2] = true;
            if (!d.dh_g && !SG_ABL(d, 0x20u)) rk[k2] = atomicAdd(&d.deg[SG_DEG_IDX(f[k2], oq & (SG_DEG_REP - 1))], 1u);   // arrival order inside the row's replica (dh_g: k2_deg_hist ranks the edges instead, no device atomic)
        }
        // (everything that does not need the returned rank first: the device atomics' round trip passes under these stores)
#pragma unroll
        for (int k2 = 0; k2 < SPT; k2++) {
            if (!live[k2]) continue;
            const u32 sl = t + (u32)k2 * NT;
            const size_t slot = (size_t)oq * d.pcap + oi[k2];
            if constexpr (WM == 2) d.wk_pos[(size_t)oq * HT + sl] = oi[k2];
            d.e_from[slot] = f[k2]; d.e_to[slot] = to[k2];
            ulonglong2* o = reinterpret_cast<ulonglong2*>(d.acc_src + slot * 4);
            if constexpr (WM == 2) {
                const bool touched = !((htouch[sl >> 5] >> (sl & 31u)) & 1u);   // (everything the records put in; the union phase marked its own)
                o[0] = make_ulonglong2(touched ? hacc[sl] : 0ull, touched ? hacc[HT + sl] : 0ull);
                o[1] = make_ulonglong2(touched ? (hacc[2 * HT + sl] | (1ull << 63)) : 0ull, touched ? hacc[3 * HT + sl] : 0ull);
            } else {
            o[0] = make_ulonglong2(hacc[sl], hacc[HT + sl]); o[1] = make_ulonglong2(hacc[2 * HT + sl], hacc[3 * HT + sl]);
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < SPT; k2++) if (live[k2] && !d.dh_g) d.e_rank[(size_t)oq * d.pcap + oi[k2]] = rk[k2];
    }
    __syncthreads();
    SG_STAMP(d, 1, 5);
    if (t == 0) {
        const u32 on = *out_n, nd = *n_drop;
        d.part_n[oq] = on < d.pcap ? on : d.pcap;
        if (nd) {                                                    // dropped after pass A had counted them as accepted
            atomicAdd(&d.ctr[C_DROPPED_CAP], (u64)nd);
            atomicAdd(&d.ctr[C_N_EVENTS], 0ull - (u64)nd);
        }
    }
}
// (the SGPR cap lets two 1024-thread workgroups share a CU — tools/occupancy_probe.hip; the uncapped build for geometries
// where a CU holds one workgroup anyway)
// (the development build's stamp / ablation code costs two registers more than the 64 that let two workgroups share a CU: there the
// allocator is told to stay within them, so that what the tools measure is the shipped kernel's occupancy)
#ifdef SG_DEV_KNOBS
#define SG_K1B_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define SG_K1B_OCC
#endif
template <int U, int SPT, bool PACK, int WM = 0> __global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(72))) SG_K1B_OCC void k1b_stream_merge(Dev d) { k1b8_body<U, SPT, PACK, WM>(d); }
template <int U, int SPT, bool PACK, int WM = 0> __global__ __launch_bounds__(1024) void k1b_stream_merge_wide(Dev d) { k1b8_body<U, SPT, PACK, WM>(d); }
