// sg_kw.h — warm and delta windows: kw_capture, kw_compact (the kept CSR -> the window's CSR), the merge of new edges
// Part of the kernel translation unit: included by sg_kernels.h (which holds the shared helpers), in this order.
#pragma once

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
