// sg_k2.h — K2 csr_build: window bookkeeping (kc_prepare), the rebuild chain (row pointers, scatter, row sort), variant 1's table compaction, the probes
// Part of the kernel translation unit: included by sg_kernels.h (which holds the shared helpers), in this order.
#pragma once

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
    // (a') pass B's order: the partitions by the records they held in the window before, largest first.  A window's keys are Zipf-distributed and
    // so are its partitions (0.7 .. 1.5 x the mean at C3); pass B runs two rounds of workgroups per CU, and a large partition that starts late
    // is the launch's tail.  Measured on one box (profiles/r06_order_ab.txt): pass B 94.5-95.3 us in this order, 97.8-99.5 in block order.  Whatever
    // the counts are, the result is a permutation (within a size class the order is that of arrival: the rows do not depend on it).
    if (d.narrow && d.k1b_order) {
        // (a counting sort by 256 size classes: two LDS atomics per partition and one wave's scan — this kernel is one workgroup on the
        // window's critical path; ranking every partition against every other cost it 14 us)
        __shared__ __attribute__((aligned(16))) u32 hb[256];
        __shared__ u32 cmax;
        const u32 NP = d.np;                                         // (<= 2048: two partitions per thread at most)
        if (t < 256) hb[t] = 0;
        if (t == 0) cmax = 0;
        __syncthreads();
        u32 c[2] = {0u, 0u}, rk[2] = {0u, 0u}, cls[2] = {0u, 0u};
#pragma unroll
        for (int k = 0; k < 2; k++) { const u32 i = t + (u32)k * 1024u; if (i < NP) { c[k] = d.k1b_cnt[i]; d.k1b_cnt[i] = 0u; } }
        { const u32 m = (u32)wave_max_u64((u64)(c[0] > c[1] ? c[0] : c[1])); if (lane == 0 && m) atomicMax(&cmax, m); }
        __syncthreads();
        const float scale = 255.0f / (float)(cmax ? cmax : 1u);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const u32 i = t + (u32)k * 1024u;
            if (i < NP) { const u32 q = (u32)((float)c[k] * scale); cls[k] = 255u - (q < 255u ? q : 255u); rk[k] = atomicAdd(&hb[cls[k]], 1u); }
        }
        __syncthreads();
        if (wave == 0) {                                             // class sizes -> first positions (four classes per lane)
            const uint4 v = reinterpret_cast<const uint4*>(hb)[lane];
            const u32 sum = v.x + v.y + v.z + v.w;
            u32 incl = sum;
            incl += dpp32<0x111>(incl); incl += dpp32<0x112>(incl); incl += dpp32<0x114>(incl); incl += dpp32<0x118>(incl);
            const u32 r0 = rdlane32(incl, 15), r1 = rdlane32(incl, 31), r2 = rdlane32(incl, 47);
            incl += (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
            const u32 ex = incl - sum;
            reinterpret_cast<uint4*>(hb)[lane] = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) { const u32 i = t + (u32)k * 1024u; if (i < NP) d.k1b_order[hb[cls[k]] + rk[k]] = i; }
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
            if (delta) d.ctr[C_DELTA_N] = E;                         // (the warm pass B only said "there are some")
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
