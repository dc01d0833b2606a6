// sg_k1_team.h — K1 pass A: k1a_team_partition (round 4; one group per tile and sixteen waves since round 6).  Included by sg_kernels.h
// behind sg_k1_narrow.h (whose record format, pieces, cache and emit_* helpers it shares).
//
// Same job as k1a_tile_partition (extractAddressPair + setFromToV2 + ReverseDirection + the per-request PersistRequest,
// aggregator/data.go:1760-1767, 827-870; datastore/dto.go:226-231; backend.go:819-847), built around what the counters and the
// issue-rate probe (profiles/r04_a_rate_probe.txt) say about it:
//
//   * its waves issue in ~20 % of their cycles and are parked at s_waitcnt / s_barrier for more than half; one wave alone issues an integer
//     VALU instruction per 5 cycles, four waves per SIMD one per 1.25; the LDS serves a random ds_read_b64 / returning ds_add_u32 in
//     4-5 cycles per wave-instruction.  So:
//   * TWO TEAMS of eight waves per workgroup, each sorting its own tiles (own tile, run counters, offsets; software team barrier on an
//     LDS counter — gfx950 has one hardware barrier per workgroup).  While one team waits for its events, scans or copies runs out, the
//     other one joins.  They share the join tables, the edge cache and the piece counters (a tile's runs get their piece positions by
//     returning LDS adds), so pass B sees the same 256 pieces per partition;
//   * the join / key / cache probe of a PAIR of events is batched and branch-free (level 1 of all four addresses, level 2, cache
//     buckets, run ranks: four dependent LDS round trips per pair), the rank taken by an unconditional returning add of 0 or 1;
//     only cache claims (a few hundred per launch), cache folds and rare events branch, each behind one wave ballot;
//   * ONE group of four events per thread and tile.  Round 4 kept two groups in flight (eight parked records per thread: 168
//     registers, 768 threads = twelve waves per CU, 6144-event tiles — a team drew three or four of them per launch); with one group
//     the kernel fits 128 registers without a spill, 1024 threads = sixteen waves, and 2048-event tiles are handed out nine or ten
//     per team (round 6, same box: 134.0 -> 127.2 us at C3).  Measured beside it and rejected: TWO workgroups per CU of 768 / 640 /
//     512 threads (80 / 96 / 128 registers, 512 pieces per partition: pass A 141-222 us, pass B +35 us for the 512 headers; level 2
//     of the join read from global memory instead of a second LDS copy: 187 us) — docs/HISTORY.md.
// Integer adds / max only: bit-exact whatever the order (the pass-A kernels run the same parity tests).
#pragma once

#define K1M_OOB     0x80000000u   // buffer offset of a lane without work (out of range of every buffer the kernel addresses)
#define K1M_RARE    0xFFFFFFFEu   // parked-record tag: a rare event, joined by the general path at the end of its tile
// LDS besides the cache and the join tables: piece counters + per team 4 counter arrays, statistics + barrier words, tile(s) of 4 records per thread (+ trash words)
#define K1M_LDS_FIXED(np, teams, nt) ((size_t)(np) * 4 * (2 + 4 * (teams)) + 128 + ((size_t)(nt) * 4 + 4) * 8)

// tiles of a launch of n events (the kernel's own arithmetic, for the host's ticket accounting)
static inline unsigned long long k1m_tiles(unsigned long long n, unsigned nwg, unsigned teams, unsigned nt) {
    const unsigned long long units = (unsigned long long)teams * nwg, tt = nt / teams, per = (n + units - 1) / units;
    const unsigned long long grp = per >= 4 * tt ? 4 * tt : (per + tt - 1) / tt * tt;
    return (n + grp - 1) / grp;                                      // a tile = one group
}
// L2M: level 2 of the join 0 = read from global memory, 1 = staged in LDS as u32, 2 = staged as u16 entries kind << 14 | id
// TEAMS: 2 = two teams (software team barriers), 1 = one team (the hardware barrier; same code otherwise: the A/B).
// NT: threads per workgroup: 1024 (four waves per SIMD, 128 registers per lane; the kernel uses 115, no scratch).
// NPB: log2 of the partition count, a compile-time constant: every counter array then sits at a constant LDS offset and the scan is
// straight-line code (with a run-time count the scan's quad loop branched per quad and its piece bases were spilled to scratch — whose
// reload waits for vmcnt(0), i.e. for the previous tile's copy-out stores to reach memory: 3 us per scan).
template <int L2M, bool SHARDED, int TEAMS, int NT, int NPB>
__global__ __launch_bounds__(NT) void k1a_team_partition(Dev d, const sg_event* __restrict__ ev, u64 n) {
    constexpr u32 K1M_THREADS = NT;
    constexpr u32 K1M_TILE_ALL = 4 * NT;                             // records in the workgroup's tile(s)
    constexpr u32 K1M_TT = K1M_THREADS / TEAMS;                      // threads per team
    constexpr u32 K1M_TW = K1M_TT / 64;                              // waves per team
    constexpr u32 K1M_GROUP = 4 * K1M_TT;                            // events per group at most: four per thread of a team
    constexpr u32 K1M_TS = K1M_TILE_ALL / TEAMS;                     // records per team tile: one group
    constexpr u32 K1M_NR = 4;                                        // records a thread parks per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 CT = d.k1a_ct;
    constexpr u32 NP = 1u << NPB;
    // LDS: piece counters | per-team counters | statistics | barrier words | tile(s) — all at constant offsets — then the join tables and the edge cache
    u32* fcn = reinterpret_cast<u32*>(smem);                         // [np] narrow records in piece (p, this workgroup): returning adds   (shared by the teams)
    u32* fcw = fcn + NP;                                             // [np] wide singles | aggregates << 16
    u32* tcnt = fcw + NP;                                            // [TEAMS][4][np]: run lengths (two sets, alternating tiles) | run offsets | piece base of the run
    u64* red = reinterpret_cast<u64*>(tcnt + 4 * TEAMS * NP);        // [8] workgroup statistics (WS_* order)
    u32* bar = reinterpret_cast<u32*>(red + 8);                      // [16] team barrier counters (team k: word 8 k)
    u64* tiles = reinterpret_cast<u64*>(bar + 16);                   // [TEAMS][K1M_TS + 2]: a team's tile, then its trash word (+ pad: 16-byte alignment)
    uint4* jl = reinterpret_cast<uint4*>(tiles + K1M_TILE_ALL + 4);  // LDS copy of the join blob: jl1 | jl2 (L2M != 0), d.jstage_bytes (a multiple of 16)
    const u32 jl1_bytes = (d.jl1mask + 1) * 8u;                      // the blob's LDS footprint: level 1 as it is, level 2 (if staged) as u32 or packed to u16 entries
    const u32 jlds = L2M == 2 ? jl1_bytes + ((d.jstage_bytes - jl1_bytes) >> 1) : d.jstage_bytes;
    u64* ckey = reinterpret_cast<u64*>(reinterpret_cast<unsigned char*>(jl) + ((jlds + 15u) & ~15u));   // [CT] mixed keys     (shared by the teams)
    u64* cacc = ckey + CT;                                           // [CT][4]
    const u64* l1 = reinterpret_cast<const u64*>(jl);
    const u32* l2 = L2M == 1 ? reinterpret_cast<const u32*>(l1 + d.jl1mask + 1) : d.jl2;
    const unsigned short* l2h = reinterpret_cast<const unsigned short*>(l1 + d.jl1mask + 1);      // L2M == 2
    const u32 w = blockIdx.x, t = threadIdx.x, lane = t & 63u;
    // (wave-uniform, and told so: the team's LDS arrays are then scalar base addresses instead of per-lane pointers)
    const u32 wave_all = (u32)__builtin_amdgcn_readfirstlane((int)(t >> 6)), team = TEAMS == 2 ? (wave_all >= K1M_TW ? 1u : 0u) : 0u, wv = wave_all - team * K1M_TW;
    const u32 tt = t - team * K1M_TT;
    u32* bcnt = tcnt + team * 4 * NP;                                // this team's [2][np] run lengths
    u32* boff = bcnt + 2 * NP;                                       // [np] run offsets inside the tile
    u32* pbase = boff + NP;                                          // [np] first piece position of the tile's run
    u64* tile = tiles + team * (K1M_TS + 2);
    u32* mybar = bar + 8 * team;
    const uint4* __restrict__ pe = reinterpret_cast<const uint4*>(ev);
    // The batch is cut into groups of g <= 2048 events (a multiple of 512: up to four events per thread of a team); a tile = one
    // group; tile j belongs to unit j % (TEAMS nwg), unit = team * nwg + workgroup — so that at any moment the 512 units read one
    // contiguous stretch of the batch and a small batch still reaches every workgroup (the pieces of a window fed by many small batches fill evenly).
    const u32 units = (u32)TEAMS * d.nwg;
    const u64 per = (n + units - 1) / units;
    const u32 grp = per >= K1M_GROUP ? K1M_GROUP : (u32)((per + K1M_TT - 1) / K1M_TT * K1M_TT);
    const u64 ngroup = (n + grp - 1) / grp, ntile = ngroup;              // a tile = one group
    // (rotated by the tiles of the window's earlier launches: many small batches must not all land on the pieces of the first workgroups)
    const u32 rot = d.k1a_rot % units;
    const u32 unit = (team * d.nwg + w + units - rot) % units, unit_other = ((1u - team) * d.nwg + w + units - rot) % units;
    const bool first = d.batch_state == 1u;                          // first batch of the window: the headers are zero by definition
    if ((u64)unit >= ntile && (TEAMS == 1 || (u64)unit_other >= ntile)) {   // no share of this batch (neither team): pieces and statistics stay as they are,
        if (first) for (u32 p = t; p < NP; p += K1M_THREADS) d.hdr8[(size_t)p * d.nwg + w] = make_uint2(0u, 0u);   // but stale headers must go
        return;
    }
    SG_STAMP(d, 0, 0);
    // Everything rare (the general path, the overflow list, the cache flush) reads its share of the ~100 Dev fields from the kernel
    // argument segment WHERE it runs (Dev is argument 0; the pointer is made opaque, so the scalar loads cannot be hoisted to the kernel's
    // entry) — held in SGPRs from the entry on they were most of ~230 live scalars, 120 of them spilled to VGPR lanes inside the tile loop.
    auto kd = [&]() -> const Dev& {
        const __attribute__((address_space(4))) Dev* kp = (const __attribute__((address_space(4))) Dev*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return *(const Dev*)kp;
    };
    const bool clk_me = w == 0 && t == 0;                            // the shader clock this launch ran at (sg_clock_probe)
    const u64 clk_c0 = clk_me ? __builtin_readcyclecounter() : 0ull, clk_r0 = clk_me ? wall_clock64() : 0ull;
    // statistics in registers: accepted events, their time-stamp range, drops for a non-pod source, largest label; what the general path and the
    // overflow paths count goes to the workgroup's LDS line where it happens
    u32 st_acc = 0, st_dsrc = 0, st_maxlabel = 0, st_misr = 0; u64 st_tmin = ~0ull, st_tmax = 0;
    auto lflush = [&](const K1Local& x) {
        if (x.acc) { atomicAdd(&red[WS_ACCEPTED], (u64)x.acc); atomicMin(&red[WS_TMIN], x.tmin); atomicMax(&red[WS_TMAX], x.tmax); }
        if (x.lost) atomicAdd(&red[WS_PAD], (u64)x.lost);
        if (x.maxlabel) atomicMax(&red[WS_MAXLABEL], (u64)x.maxlabel);
        if (x.dsrc) atomicAdd(&red[WS_DROPPED_SRC], (u64)x.dsrc);
        if (x.dcap) atomicAdd(&red[WS_DROPPED_CAP], (u64)x.dcap);
        if (x.misr) atomicAdd(&red[WS_MISROUTED], (u64)x.misr);
    };
#define K1M_LNEW(L) K1Local L; L.tmin = ~0ull; L.tmax = 0; L.maxlabel = L.dsrc = L.dcap = L.misr = L.acc = L.lost = 0
    const u32 nb = d.nb, nbmask = (1u << nb) - 1u, pshift = nb - d.pb, rbmask = (1u << d.rb) - 1u, bmask = CT / 2 - 1, jm = d.jl1mask;
    const u32 max_known = d.max_known, max_labels = d.max_labels, sn = d.sn;
    const bool ck_any = d.ck_n != 0;
#define K1M_WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory")   /* all 16 waves; LDS-only: does not drain the global stores */
    // The team barrier: a wave's own LDS operations are complete (lgkmcnt(0)), one lane adds 1 to the team's counter, every wave
    // polls it (one broadcast ds_read per trip) until all eight of this round have arrived.  The counter only grows.
    u32 bar_target = 0;
    auto team_barrier = [&]() {
        if constexpr (TEAMS == 1) { K1M_WG_BARRIER(); return; }
        bar_target += K1M_TW;
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        if (lane == 0) __hip_atomic_fetch_add(mybar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while ((int)(lds_fresh_u32(mybar) - bar_target) < 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" : : : "memory");
    };

    // cache fold of one accepted event into a slot it owns
    auto cache_add = [&](u32 slot, u64 dur, u32 err) {
        u64 ssq;
        if ((dur >> 32) == 0) { const u32 us = div1000_u32((u32)dur); ssq = (u64)us * (u64)us; }
        else { const u64 us = dur / 1000ull; ssq = us * us; }
        atomicAdd(&cacc[slot * 4], 1ull | ((u64)err << 32)); atomicAdd(&cacc[slot * 4 + 1], dur);
        atomicMax(&cacc[slot * 4 + 2], dur); atomicAdd(&cacc[slot * 4 + 3], ssq);
    };
    // The general path (rare events: open connections, raw-IP outbound destinations, IPs in both maps or in the residual
    // cuckoo table, durations of 2^32 ns and more, labels out of range): the full join on the global tables, at the END of the event's tile
    // (where nothing of the batched fast path is live); the event is read again — it is rare and in L2.  Its record does not join the tile:
    // a narrow record takes its piece position from the shared counter, like the cache flush's.
    auto general = [&](const u64 idx) {
        const Dev& d = kd();                                         // (shadows the kernel's copy: see kd)
        const uint4* q2 = pe + 2 * idx;
        const uint4 xa = q2[0], xb = q2[1];
        K1Ev e;
        K1M_LNEW(L);
        const bool ok = k1_resolve(d, xa, xb, L, e);
        if (!ok) { lflush(L); return; }
        u32 Lm, Rm;
        sg_kmix(ci_of_ref(d, (u32)(e.key >> 32)), ci_of_ref(d, (u32)e.key), nbmask, &Lm, &Rm);
        const u32 part = Lm >> pshift;
        const u64 mk = ((u64)Lm << nb) | Rm;
        if (e.alive) { emit_wide(d, fcw, w, part, mk, 0ull, 0u, 1u, L); lflush(L); return; }
        const u32 bkt = Rm & bmask;
        const int slot = cache_claim(ckey, bkt, mk, lds_fresh_u64(&ckey[2 * bkt]), lds_fresh_u64(&ckey[2 * bkt + 1]));
        if (slot >= 0) cache_add((u32)slot, e.dur, e.err);
        else if (e.dur >> 32) emit_wide(d, fcw, w, part, mk, e.dur, e.err, 0u, L);
        else emit_narrow_direct(d, fcn, w, part, (u32)mk & rbmask, (u32)e.dur, e.err, L);
        lflush(L);
    };
    constexpr u32 KSH = L2M == 2 ? 14u : 30u, IDM = L2M == 2 ? 0x3FFFu : 0x3FFFFFFFu;   // kind shift / id mask of a level-2 entry as this build reads it

#define K1M_F_ACC 1u
#define K1M_F_RARE 2u
#define K1M_F_ERR 4u
    // ---- a PAIR of events: the batched fast path, in two halves -----------------------------------------------------
    // front: both joins (two-level block table in LDS: level 1 of all four addresses — eight independent ds_read_b64 —, then level 2),
    //        data.go:827-870 as selects, error rule, time stamps, key mix.  Leaves per event: the mixed key halves, the duration's low word
    //        and three flag bits (a rare event is re-read from memory by the general path).
    // back:  cache bucket reads, claims (one ballot), the run ranks (unconditional returning adds of 0 or 1), folds, parked records.
    // (round 4 batched four events per step — four dependent LDS round trips per group of four — at 168 registers and twelve waves per
    // CU; two per step fit 128 registers without a spill: sixteen waves, and the round trips a pair leaves exposed are another wave's
    // issue slots.)  i0: index of the pair's first event in the group (0 or 2).
    auto front2 = [&](const u32 tq, const u32 i0, const u32 gcount, const v4u_t A0, const v4u_t B0, const v4u_t A1, const v4u_t B1,
                      u32 (&Lm)[2], u32 (&Rm)[2], u32 (&dur)[2], u32 (&fl)[2]) {
        const v4u_t A[2] = {A0, A1}, B[2] = {B0, B1};
        u32 vs[2], vd[2];
        {
            u64 s1[2], s2[2], d1[2], d2[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const u32 b = A[i].x >> 8, c = A[i].y >> 8;
                s1[i] = l1[(__umul24(b, SG_JL1_K1) >> 9) & jm]; s2[i] = l1[(__umul24(b, SG_JL1_K2) >> 11) & jm];
                d1[i] = l1[(__umul24(c, SG_JL1_K1) >> 9) & jm]; d2[i] = l1[(__umul24(c, SG_JL1_K2) >> 11) & jm];
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const u32 b = A[i].x >> 8, c = A[i].y >> 8;
                const u32 blk = (u32)s1[i] == b ? (u32)(s1[i] >> 32) : ((u32)s2[i] == b ? (u32)(s2[i] >> 32) : 0u);
                const u32 blkd = (u32)d1[i] == c ? (u32)(d1[i] >> 32) : ((u32)d2[i] == c ? (u32)(d2[i] >> 32) : 0u);
                const u32 ix = (blk << 8) | (A[i].x & 255u), ixd = (blkd << 8) | (A[i].y & 255u);
                vs[i] = L2M == 2 ? (u32)l2h[ix] : l2[ix];
                vd[i] = L2M == 2 ? (u32)l2h[ixd] : l2[ixd];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const bool inr = tq + (i0 + (u32)i) * K1M_TT < gcount;
            const u32 flags = A[i].w >> 24, label = A[i].z;
            const u32 ks = vs[i] >> KSH, kd = vd[i] >> KSH;
            bool r = ((flags & SG_EV_ALIVE) != 0) | (ks == 3u) | (kd == 3u) | (B[i].y != 0u) |
                     ((kd == 0u) & ((label == 0u) | (label > max_labels))) | (ck_any & ((vs[i] == 0u) | (vd[i] == 0u)));
            r &= inr;
            const bool fastv = inr & !r;
            bool a = fastv & (ks == 1u);                             /* data.go:829-832: the source must be a pod */
            st_dsrc += (fastv & (ks != 1u)) ? 1u : 0u;
            u32 cf = vs[i] & IDM;
            u32 ct = kd ? (vd[i] & IDM) : (max_known + label - 1u);  /* service / pod id, else Host label (:840-854) */
            { const u32 lb = (a & (kd == 0u)) ? label : 0u; st_maxlabel = lb > st_maxlabel ? lb : st_maxlabel; }
            if (flags & SG_EV_REVERSE) { const u32 x_ = cf; cf = ct; ct = x_; }      /* dto.go:226-231 */
            if (SHARDED) { const bool mine = (owner_hash_ref(ref_of_ci(d, cf)) % d.world) == d.rank; st_misr += (a & !mine) ? 1u : 0u; a &= mine; }
            const u32 status = A[i].w & 0xFFFFu, proto = (A[i].w >> 16) & 0xFFu;
            const u32 err = is_error(proto, status);
            const u64 wt = (u64)B[i].z | ((u64)B[i].w << 32);
            st_acc += a ? 1u : 0u;
            st_tmin = (a && wt < st_tmin) ? wt : st_tmin; st_tmax = (a && wt > st_tmax) ? wt : st_tmax;
            sg_kmix(cf & nbmask, ct & nbmask, nbmask, &Lm[i], &Rm[i]);
            dur[i] = B[i].x;
            fl[i] = (a ? K1M_F_ACC : 0u) | (r ? K1M_F_RARE : 0u) | (err ? K1M_F_ERR : 0u);
        }
    };
    auto back2 = [&](const u32 (&Lm)[2], const u32 (&Rm)[2], const u32 (&dur)[2], const u32 (&fl)[2], u32* bc, u32* lo, u32* hi, u32* pr) {
        ulonglong2 kk[2]; u64 mk[2];
#pragma unroll
        for (int i = 0; i < 2; i++) { mk[i] = ((u64)Lm[i] << nb) | Rm[i]; kk[i] = reinterpret_cast<const ulonglong2*>(ckey)[Rm[i] & bmask]; }
        int slot[2]; bool claim = false;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const u32 bk = Rm[i] & bmask;
            slot[i] = kk[i].x == mk[i] ? (int)(2u * bk) : (kk[i].y == mk[i] ? (int)(2u * bk + 1u) : -1);
            claim |= ((fl[i] & K1M_F_ACC) != 0) & (slot[i] < 0) & ((kk[i].x == SG_EKEY_EMPTY) | (kk[i].y == SG_EKEY_EMPTY));
        }
        if (__builtin_amdgcn_ballot_w64(claim)) {
#pragma unroll
            for (int i = 0; i < 2; i++)
                if ((fl[i] & K1M_F_ACC) && slot[i] < 0 && (kk[i].x == SG_EKEY_EMPTY || kk[i].y == SG_EKEY_EMPTY)) slot[i] = cache_claim(ckey, Rm[i] & bmask, mk[i], kk[i].x, kk[i].y);
        }
        u32 rank[2];
#pragma unroll
        for (int i = 0; i < 2; i++) rank[i] = atomicAdd(&bc[Lm[i] >> pshift], ((fl[i] & K1M_F_ACC) && slot[i] < 0) ? 1u : 0u);
#pragma unroll
        for (int i = 0; i < 2; i++) if ((fl[i] & K1M_F_ACC) && slot[i] >= 0) cache_add((u32)slot[i], (u64)dur[i], (fl[i] >> 2) & 1u);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const bool travel = (fl[i] & K1M_F_ACC) && slot[i] < 0;
            lo[i] = dur[i]; hi[i] = ((u32)mk[i] & rbmask) | ((fl[i] & K1M_F_ERR) ? 0x80000000u : 0u);
            pr[i] = travel ? ((Lm[i] >> pshift) | (rank[i] << K1T_RANK_SHIFT)) : ((fl[i] & K1M_F_RARE) ? K1M_RARE : K1T_NONE);
        }
    };

    {   // prologue (all 16 waves): counters, cache, statistics, the join blob
        for (u32 p = t; p < NP; p += K1M_THREADS) {
            uint2 h = make_uint2(0u, 0u);
            if (!first) h = d.hdr8[(size_t)p * d.nwg + w];
            fcn[p] = h.x; fcw[p] = h.y;
        }
        for (u32 k = t; k < 4 * TEAMS * NP; k += K1M_THREADS) tcnt[k] = 0u;
        for (u32 k = t; k < CT; k += K1M_THREADS) ckey[k] = SG_EKEY_EMPTY;
        for (u32 k = t; k < CT * 4; k += K1M_THREADS) cacc[k] = 0;
        if (t < 8) red[t] = t == WS_TMIN ? ~0ull : 0ull;
        if (t < 16) bar[t] = 0u;
        // the join blob: six 16-byte loads per lane at most, issued and waited for in ONE asm statement (no code may sit between a
        // hand-issued load and the wait that names its registers)
        v4u_t jb0, jb1, jb2, jb3, jb4, jb5;
        static_assert(K1A_NJ == 6, "written out for 6 blob words per lane");
        const u32 n16 = d.jstage_bytes >> 4, n1 = (d.jl1mask + 1) >> 1;   // 16-byte words to stage; of them level 1 (always there)
        const uint4* g1 = reinterpret_cast<const uint4*>(d.jl1); const uint4* g2 = reinterpret_cast<const uint4*>(d.jl2) - n1;
#define K1M_JIDX(k) ((t + (k) * K1M_THREADS) < n16 ? (t + (k) * K1M_THREADS) : n16 - 1)
#define K1M_JSRC(k) ((K1M_JIDX(k) < n1 ? g1 : g2) + K1M_JIDX(k))
        const uint4* js0 = K1M_JSRC(0); const uint4* js1 = K1M_JSRC(1); const uint4* js2 = K1M_JSRC(2);
        const uint4* js3 = K1M_JSRC(3); const uint4* js4 = K1M_JSRC(4); const uint4* js5 = K1M_JSRC(5);
        asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %7, off\n\tglobal_load_dwordx4 %2, %8, off\n\t"
                     "global_load_dwordx4 %3, %9, off\n\tglobal_load_dwordx4 %4, %10, off\n\tglobal_load_dwordx4 %5, %11, off\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(jb0), "=&v"(jb1), "=&v"(jb2), "=&v"(jb3), "=&v"(jb4), "=&v"(jb5)
                     : "v"(js0), "v"(js1), "v"(js2), "v"(js3), "v"(js4), "v"(js5) : "memory");
#define K1M_P16(x) ((((x) >> 30) << 14) | ((x) & 0x3FFFu))
#define K1M_JST(k, r) { const u32 i_ = t + (k) * K1M_THREADS;                                                          \
            if (i_ < n16) { if (L2M == 2 && i_ >= n1) reinterpret_cast<uint2*>(jl + n1)[i_ - n1] = make_uint2(K1M_P16((r).x) | (K1M_P16((r).y) << 16), K1M_P16((r).z) | (K1M_P16((r).w) << 16)); \
                            else jl[i_] = make_uint4((r).x, (r).y, (r).z, (r).w); } }
        K1M_JST(0, jb0); K1M_JST(1, jb1); K1M_JST(2, jb2); K1M_JST(3, jb3); K1M_JST(4, jb4); K1M_JST(5, jb5);
#undef K1M_JST
#undef K1M_P16
#undef K1M_JSRC
#undef K1M_JIDX
        K1M_WG_BARRIER();
        SG_STAMP(d, 0, 1);
    }
    // PACKB: the partition number rides in the free bits [rb, rb + pb) of a parked record's high word (2 nb <= 31): the copy-out
    // is then one thread per tile position; otherwise 16 lanes walk a run.
    const bool packb = 2u * nb <= 31u;
    const u32 rb = d.rb;
    u64* const slab_w = d.slab8 + (size_t)w * d.punits;                  // piece (p, this workgroup) = slab_w + p * nwpun
    const u32 nwpun = d.nwg * d.punits;
    // (the slab as a buffer: the host takes this kernel only when the whole slab is below 4 GiB)
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(slab_w, 0, (int)(u32)(((size_t)NP * d.nwg - w) * d.punits * 8u), 0x00020000);
    // P4 of a tile: copy its runs to the pieces — adjacent lanes, adjacent addresses.  It runs at the TOP of the team's next tile, behind
    // that tile's first loads.  Legal anywhere between the team barrier behind P3 and the next one behind P1: the tile, the offsets, the
    // piece bases and this tile's run counters are not written before that.
    // (a record beyond its piece's capacity goes to the window's overflow list: a rolled loop behind the copy, so that the eight
    // unrolled copies carry one compare each and no call-sized code)
    auto ovf_record = [&](const u32 b, const u32 remv, const u32 dur, const u32 err) { const Dev& d = kd(); K1M_LNEW(L); ovf8_single(d, b, ((u64)b << d.rb) | remv, (u64)dur, err, 0u, L); lflush(L); };
    auto copy_out = [&](const u32 pc) {
        const u32* bc = bcnt + pc * NP;
        if (packb) {
            // all eight tile positions of a thread together: eight ds_read_b64 (one base register, immediate offsets), sixteen ds_read_b32
            // (piece base and run offset of each record's partition), then the stores — no branch between the LDS reads.  (tq: the thread's
            // index made opaque per call, or the compiler hoists the eight position constants out of the tile loop and SPILLS them)
            u32 tq = tt; asm volatile("" : "+v"(tq));
            const u32 total = boff[NP - 1] + bc[NP - 1];             // records in the tile
            const u32 pbm = (1u << d.pb) - 1u, strip = ~(pbm << rb); // partition bits of a parked record's high word (bit 31 = error stays)
            constexpr u32 NPOS = K1M_TS / K1M_TT;
            u64 rec[NPOS]; u32 pb_[NPOS], bo_[NPOS];
#pragma unroll
            for (u32 k = 0; k < NPOS; k++) rec[k] = tile[tq + k * K1M_TT];
#pragma unroll
            for (u32 k = 0; k < NPOS; k++) { const u32 b = ((u32)(rec[k] >> 32) >> rb) & pbm; pb_[k] = pbase[b]; bo_[k] = boff[b]; }
            u32 ovm = 0;
#pragma unroll
            for (u32 k = 0; k < NPOS; k++) {
                const u32 i = tq + k * K1M_TT, h = (u32)(rec[k] >> 32), b = (h >> rb) & pbm;
                const u32 pos = pb_[k] + (i - bo_[k]);
                const bool valid = i < total;
                if SG_ABL(d, 0x2u) { if (valid & (pos < sn)) slab_w[(u64)b * nwpun + pos] = (rec[k] & 0xFFFFFFFFull) | ((u64)(h & strip) << 32); }
                else { v2u_t dv; dv.x = (u32)rec[k]; dv.y = h & strip;           // (a lane without a record, or beyond the piece: out-of-range offset, dropped by the hardware)
                  __builtin_amdgcn_raw_buffer_store_b64(dv, slab_rsrc, (valid & (pos < sn)) ? (b * nwpun + pos) * 8u : K1M_OOB, 0, 0); }
                ovm |= (valid & (pos >= sn)) ? (1u << k) : 0u;
            }
            if (__builtin_amdgcn_ballot_w64(ovm != 0)) {
#pragma unroll 1
                for (u32 k = 0; k < NPOS; k++) if ((ovm >> k) & 1u) {
                    const u64 r = tile[tq + k * K1M_TT];
                    const u32 h = (u32)(r >> 32);
                    ovf_record((h >> rb) & pbm, h & rbmask, (u32)r, h >> 31);
                }
            }
        } else {
            const u32 bpw = NP / K1M_TW;                             // partitions whose runs a wave writes out (np >= 64; NT = 1024 only: the host does not pick 768 threads here): four per step, 16 lanes each
#pragma unroll 1
            for (u32 b4 = 0; b4 < bpw; b4 += 4) {
                const u32 b = wv * bpw + b4 + (lane >> 4), j0 = lane & 15u;
                const u32 cnt = bc[b], off = boff[b], pos0 = pbase[b];
                u64* dst = piece8(d, b, w);
#pragma unroll 1
                for (u32 j = j0; j < cnt; j += 16) {
                    const u64 rec = tile[off + j];
                    const u32 pos = pos0 + j;
                    if (pos < sn) dst[pos] = rec;
                    else ovf_record(b, (u32)(rec >> 32) & rbmask, (u32)rec, (u32)(rec >> 63));
                }
            }
        }
    };
    bool havep = false; u32 pcur = 0, cur = 0;
    u64 tk_p1 = 0, tk_wait = 0, tk_scan = 0, tk_p3 = 0, tk_b3 = 0, tk_p4 = 0, tk_ld = 0, tk_fa = 0;   // SG_K1_PHASE_STAMPS: wave 0's clock ticks per phase (tk_ld: inside P1, waiting for the event loads; tk_fa: joining the first group)
#ifdef SG_K1_PHASE_STAMPS
    const bool stamp = SG_ABL(d, 0x100u) != 0;
#else
    constexpr bool stamp = false;                                    // (phase clocks of wave 0: build with -DSG_K1_PHASE_STAMPS; they cost a dozen registers)
#endif
    // events [first, first + count) of group g (count 0: no such group).  A launch has fewer than 2^27 events (the host splits larger
    // batches): event indices and byte offsets fit 32 bits
    const u32 n32 = (u32)n, ngroup32 = (u32)ngroup;
    auto gbounds = [&](const u32 g, u32& first, u32& count) {
        first = g * grp; count = g < ngroup32 ? (first + grp < n32 ? grp : n32 - first) : 0u;
    };
    // Tiles are handed out DYNAMICALLY: a team's first tile is its unit number, every further one a ticket from the window slot's device
    // counter (tile = units + ticket - base; the host advances the base by what a launch consumes: one ticket per tile beyond the first
    // round + one failing ticket per active team).  On static shares the teams of a launch ended 110 .. 140 us apart.  The ticket is drawn
    // by one lane at the top of the tile before (a returning device atomic issued by hand: its round trip passes under the joins) and
    // handed to the team through LDS across the tile's barriers.  (SG_ABLATE & 0x4: static shares.)
    const bool dyn = !SG_ABL(d, 0x4u);
    // (the ticket is handed on through TWO words, alternating with the tile: a wave that is late behind a tile's second barrier — rare events,
    // statistics — reads its word while the drawer, already in the next tile, writes the other one; with one word nothing ordered that read
    // before that write, and a wave that read the NEXT tile's ticket left its team for good: a hang, seen in round 6 when the write moved
    // closer to the top of the tile)
    u32* nxt2 = bar + 8 * team + 4;
    const u32 ntile32 = (u32)ntile;
    for (u32 j = unit; j < ntile32; cur ^= 1u) {
        u32* bc = bcnt + cur * NP;
        u32 lo[K1M_NR], hi[K1M_NR], pr[K1M_NR];
        // the thread's indices, opaque per tile: whatever is derived from them (tile / counter / event addresses) is then computed where it is
        // used — hoisted out of this loop as loop invariants they were SPILLED (45 dwords) and every reload's vmcnt(0) also waited for the
        // event loads in flight
        u32 ttl = tt, lanel = lane; asm volatile("" : "+v"(ttl), "+v"(lanel));
        const u64 tk0 = stamp ? wall_clock64() : 0ull;
        {   // P1: one group of up to four events per thread.
            // The previous tile's records leave FIRST, then the eight loads and the ticket go out and are waited for with vmcnt(0): a
            // vmcnt(N) that lets the copy's stores fly behind the loads would have to assume that loads and stores retire in order with
            // each other and that a store whose lanes are all out of range still counts — measured: every tile but a workgroup's first read
            // its events before they had landed.  The latency this team no longer hides is the other team's issue time.  (Keeping a set of
            // loads in flight ACROSS the tile loop's back edge was tried in round 4: the compiler spills the in-flight registers there —
            // tools/check_asm_loads.py.)
            if (havep) { const u64 tq = stamp ? wall_clock64() : 0ull; copy_out(pcur); if (stamp) tk_p4 += wall_clock64() - tq; }
            u32 f0, c0;
            gbounds(j, f0, c0);
            v4u_t eaa0, eba0, eaa1, eba1, eaa2, eba2, eaa3, eba3;
            {   // scalar base + one 32-bit offset register per event, both halves from it (a lane without work reads the batch's first event
                // and ignores it by its in-range flag)
                const u32 x0 = ttl, x1 = x0 + K1M_TT, x2 = x1 + K1M_TT, x3 = x2 + K1M_TT, fb = f0 * 32u;
                const u32 o0 = x0 < c0 ? fb + x0 * 32u : 0u, o1 = x1 < c0 ? fb + x1 * 32u : 0u, o2 = x2 < c0 ? fb + x2 * 32u : 0u, o3 = x3 < c0 ? fb + x3 * 32u : 0u;
#define K1M_LD2(A, B, vo) asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16" : "=&v"(A), "=&v"(B) : "v"(vo), "s"(pe) : "memory")
                K1M_LD2(eaa0, eba0, o0); K1M_LD2(eaa1, eba1, o1); K1M_LD2(eaa2, eba2, o2); K1M_LD2(eaa3, eba3, o3);
#undef K1M_LD2
            }
            u32 tkv = 0;
            const bool drawer = dyn && wv == 0 && lanel == 0;
            if (drawer) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(tkv) : "v"(d.k1a_ticket), "v"(1u) : "memory");
            const u64 tl0 = stamp ? wall_clock64() : 0ull;
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(eaa0), "+v"(eba0), "+v"(eaa1), "+v"(eba1), "+v"(eaa2), "+v"(eba2), "+v"(eaa3), "+v"(eba3), "+v"(tkv) : : "memory");
            if (stamp) tk_ld += wall_clock64() - tl0;
            if (drawer) { asm volatile("" : "+v"(tkv)); nxt2[cur] = tkv - d.k1a_ticket_base + units; }
            {
                u32 Lm[2], Rm[2], du[2], fl[2];
                front2(ttl, 0u, c0, eaa0, eba0, eaa1, eba1, Lm, Rm, du, fl);
                back2(Lm, Rm, du, fl, bc, lo, hi, pr);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                u32 Lm[2], Rm[2], du[2], fl[2];
                front2(ttl, 2u, c0, eaa2, eba2, eaa3, eba3, Lm, Rm, du, fl);
                back2(Lm, Rm, du, fl, bc, lo + 2, hi + 2, pr + 2);
            }
        }
        const u64 tk1 = stamp ? wall_clock64() : 0ull;
        team_barrier();
        const u64 tk2 = stamp ? wall_clock64() : 0ull;
        tk_p1 += tk1 - tk0; tk_wait += tk2 - tk1;
        {   // P2: exclusive scan of the run lengths by EVERY wave of the team (eight identical scans cost less than a barrier behind one).
            // Lane l owns the np / 64 consecutive partitions from l * np / 64; the wave that owns a partition (wave = partition / (np / 8)) also
            // takes the run's piece positions (a returning add on the piece counter the two teams share) and re-arms the other set of run counters.
            constexpr u32 pl = NP >> 6; const u32 b0 = lanel * pl;
            const bool own = (lanel % K1M_TW) == wv;                 // (any disjoint cover of the 64 lanes' partition blocks by the team's waves)
            u32* bprev = bcnt + (cur ^ 1u) * NP;                     // tile k - 1's set: read for the last time in its copy-out, every wave of the team is past that
            if constexpr ((pl & 3u) == 0 && pl <= 16) {
                constexpr u32 nq = pl >> 2;                          // 1, 2 or 4 quads per lane (np = 256, 512, 1024)
                uint4 cq[4];
                u32 s = 0;
#pragma unroll
                for (u32 k = 0; k < 4; k++) { cq[k] = k < nq ? reinterpret_cast<const uint4*>(bc + b0)[k] : make_uint4(0u, 0u, 0u, 0u); s += cq[k].x + cq[k].y + cq[k].z + cq[k].w; }
                u32 incl = s;                                        // inclusive scan over the 64 lanes: DPP row_shr 1, 2, 4, 8 (zero fill), then the row totals
                incl += dpp32<0x111>(incl); incl += dpp32<0x112>(incl); incl += dpp32<0x114>(incl); incl += dpp32<0x118>(incl);
                const u32 r0 = rdlane32(incl, 15), r1 = rdlane32(incl, 31), r2 = rdlane32(incl, 47);
                incl += (lanel >= 16 ? r0 : 0u) + (lanel >= 32 ? r1 : 0u) + (lanel >= 48 ? r2 : 0u);
                u32 run = incl - s;
                // (the owner's returning adds first, ALL in flight together — written behind each quad's offsets they were issued and
                // waited for one by one: eight LDS round trips per tile in the scan of every owner lane)
                uint4 pbv[4];
                if (own) {
#pragma unroll
                    for (u32 k = 0; k < 4; k++) if (k < nq) {
                        pbv[k].x = atomicAdd(&fcn[b0 + 4 * k], cq[k].x); pbv[k].y = atomicAdd(&fcn[b0 + 4 * k + 1], cq[k].y);
                        pbv[k].z = atomicAdd(&fcn[b0 + 4 * k + 2], cq[k].z); pbv[k].w = atomicAdd(&fcn[b0 + 4 * k + 3], cq[k].w);
                    }
                }
#pragma unroll
                for (u32 k = 0; k < 4; k++) if (k < nq) {
                    uint4 o; o.x = run; run += cq[k].x; o.y = run; run += cq[k].y; o.z = run; run += cq[k].z; o.w = run; run += cq[k].w;
                    reinterpret_cast<uint4*>(boff + b0)[k] = o;
                }
                if (own) {
#pragma unroll
                    for (u32 k = 0; k < 4; k++) if (k < nq) {
                        reinterpret_cast<uint4*>(pbase + b0)[k] = pbv[k];
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
                    if (own) { pbase[b0 + k] = atomicAdd(&fcn[b0 + k], c); bprev[b0 + k] = 0u; }
                }
            }
        }
        const u64 tk3 = stamp ? wall_clock64() : 0ull;
        // P3: every thread drops its records at offset + rank (PACKB: with the partition number in the free bits of the high word): the eight
        // offset reads together, then eight unconditional stores — a slot without a travelling record writes to the trash word behind the tile
        {
            u32 of_[K1M_NR];
#pragma unroll
            for (int i = 0; i < (int)K1M_NR; i++) of_[i] = boff[pr[i] < K1M_RARE ? (pr[i] & ((1u << K1T_RANK_SHIFT) - 1u)) : 0u];
#pragma unroll
            for (int i = 0; i < (int)K1M_NR; i++) {
                const u32 pt_ = pr[i] & ((1u << K1T_RANK_SHIFT) - 1u);
                const u32 at = pr[i] < K1M_RARE ? of_[i] + (pr[i] >> K1T_RANK_SHIFT) : K1M_TS;
                tile[at] = (u64)lo[i] | ((u64)(hi[i] | (packb ? pt_ << rb : 0u)) << 32);
            }
        }
        const u64 tk4 = stamp ? wall_clock64() : 0ull;
        team_barrier();
        const u64 tk5 = stamp ? wall_clock64() : 0ull;
        {   // the tile's rare events: the general path, one event at a time (nothing of P1 is live here)
            u32 rmask = 0;
#pragma unroll
            for (int i = 0; i < (int)K1M_NR; i++) rmask |= (pr[i] == K1M_RARE) ? (1u << i) : 0u;
            if (__builtin_amdgcn_ballot_w64(rmask != 0)) {
                const u64 e0 = (u64)j * grp + ttl;                   // (a rare event is in range: its group exists)
#pragma unroll 1
                for (u32 k = 0; k < K1M_NR; k++) if ((rmask >> k) & 1u) general(e0 + (u64)k * K1M_TT);
            }
        }
        {
            // the lanes' statistics leave per tile (wave reduce -> the workgroup's LDS line): eight registers that are not carried around the loop
            const u64 tmin = wave_min_u64(st_tmin), tmax = wave_max_u64(st_tmax);
            const u32 ac = wave_sum_u32(st_acc);
            if (lanel == 0 && ac) { atomicMin(&red[WS_TMIN], tmin); atomicMax(&red[WS_TMAX], tmax); atomicAdd(&red[WS_ACCEPTED], (u64)ac); }
            if (__builtin_amdgcn_ballot_w64((st_dsrc | st_misr | st_maxlabel) != 0u)) {
                if (st_dsrc) atomicAdd(&red[WS_DROPPED_SRC], (u64)st_dsrc);
                if (st_misr) atomicAdd(&red[WS_MISROUTED], (u64)st_misr);
                if (st_maxlabel) atomicMax(&red[WS_MAXLABEL], (u64)st_maxlabel);
            }
            st_acc = st_dsrc = st_maxlabel = st_misr = 0; st_tmin = ~0ull; st_tmax = 0;
        }
        havep = true; pcur = cur;
        if (stamp) { tk_scan += tk3 - tk2; tk_p3 += tk4 - tk3; tk_b3 += tk5 - tk4; }
        j = dyn ? (u32)__builtin_amdgcn_readfirstlane((int)lds_fresh_u32(nxt2 + cur)) : j + units;   // (written before this tile's first barrier, read behind its second)
    }
    if (havep) copy_out(pcur);                                       // the last tile's runs
    SG_STAMP(d, 0, 3);
    if (stamp && tt == 0 && blockIdx.x < 2048) { u64* g = d.dbg + ((size_t)2 * 4096 + blockIdx.x * 2 + team) * 8; g[0] = tk_p1; g[1] = tk_wait; g[2] = tk_scan; g[3] = tk_p3; g[4] = tk_b3; g[5] = tk_p4; g[6] = tk_ld; g[7] = tk_fa; }
    K1M_WG_BARRIER();
    SG_STAMP(d, 0, 4);
    // flush the cache: a key seen once leaves as a single record, the others as aggregates
    {
    const Dev& d = kd();                                             // (the epilogue's Dev fields are loaded here, not held across the tile loop)
    K1M_LNEW(L);
    for (u32 s = t; s < CT; s += K1M_THREADS) {
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
        const u32 ac = wave_sum_u32(st_acc), ds = wave_sum_u32(st_dsrc), mr = wave_sum_u32(st_misr);
        const u32 ml = (u32)wave_max_u64((u64)st_maxlabel);
        if (lane == 0) {
            if (ac) { atomicMin(&red[WS_TMIN], tmin); atomicMax(&red[WS_TMAX], tmax); atomicAdd(&red[WS_ACCEPTED], (u64)ac); }
            if (ds) atomicAdd(&red[WS_DROPPED_SRC], (u64)ds);
            if (mr) atomicAdd(&red[WS_MISROUTED], (u64)mr);
            if (ml) atomicMax(&red[WS_MAXLABEL], (u64)ml);
        }
    }
    K1M_WG_BARRIER();
    for (u32 p = t; p < NP; p += K1M_THREADS) { const u32 c = fcn[p]; d.hdr8[(size_t)p * d.nwg + w] = make_uint2(c < sn ? c : sn, fcw[p]); }
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
    }
#undef K1M_WG_BARRIER
#undef K1M_LNEW
}
