// sg_k3.h — K3 node statistics and features, the window reset
// Part of the kernel translation unit: included by sg_kernels.h (which holds the shared helpers), in this order.
#pragma once

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
#define K3_IN_SMAX  48        // edge slices at most (48 where ranges x 48 workgroups are one round of the chip, else 32: servicegraph.hip)
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

// fin (the one-call pipelines, whose node features sum the slices' partials themselves — k3_node_features — so that k3_in_reduce is not
// launched): the launch's LAST fin workgroups do what k3_in_reduce's last ones do in the staged pipelines — finish the rows behind
// kw_compact (degree, mean / deviation, hub work items), or the block-sorted rows of an engine without the kept state.
__global__ __launch_bounds__(1024) void k3_in_part(Dev d, u32 S, u32 fin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 E = (u32)d.ctr[C_N_EDGES], N = (u32)d.ctr[C_N_NODES];
    const u32 G = gridDim.x - fin, g = blockIdx.x, t = threadIdx.x;
    if (g >= G) {
        if (d.warm) kw_finish_rows(d, (g - G) * 1024 + t, fin * 1024);
        else if (g == gridDim.x - 1) k2_split_finish(d, t, 1024);
        return;
    }
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
// S_in != 0 (the one-call pipelines): the in-side lane of a node sums the S_in slices' partials of k3_in_part itself (and leaves the sums in
// st_sum / st_max, where the staged pipelines' k3_in_reduce puts them) — one launch and a 24 MB round trip less per window.
__global__ __launch_bounds__(256) void k3_node_features(Dev d, u32 nb_nodes, u32 S_in) {
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
            u64* s = d.st_sum + (size_t)v * SG_NODE_STAT_SUM_WORDS;
            u64 dg, c, er, sm, sq, mx;
            if (S_in && side) {                                      // deg, cnt, err, sum, ssq, max of the node's in-edges: over the slices (integer: order-free)
                const u32 r = v / K3_IN_NR;
                const ulonglong2* p = reinterpret_cast<const ulonglong2*>(d.in_part + ((size_t)r * S_in * K3_IN_NR + (v - r * K3_IN_NR)) * 6);
                const size_t st = (size_t)K3_IN_NR * 3;              // (16-byte words per slice)
                dg = c = er = sm = sq = mx = 0;
                u32 sl = 0;
                for (; sl + 4 <= S_in; sl += 4) {                    // twelve independent loads in flight
                    ulonglong2 x[4][3];
#pragma unroll
                    for (int q = 0; q < 4; q++) { x[q][0] = p[(size_t)(sl + q) * st]; x[q][1] = p[(size_t)(sl + q) * st + 1]; x[q][2] = p[(size_t)(sl + q) * st + 2]; }
#pragma unroll
                    for (int q = 0; q < 4; q++) { dg += x[q][0].x; c += x[q][0].y; er += x[q][1].x; sm += x[q][1].y; sq += x[q][2].x; mx = x[q][2].y > mx ? x[q][2].y : mx; }
                }
                for (; sl < S_in; sl++) {
                    const ulonglong2 a0 = p[(size_t)sl * st], a1 = p[(size_t)sl * st + 1], a2 = p[(size_t)sl * st + 2];
                    dg += a0.x; c += a0.y; er += a1.x; sm += a1.y; sq += a2.x; mx = a2.y > mx ? a2.y : mx;
                }
                s[ST_IN_DEG] = dg; s[ST_IN_CNT] = c; s[ST_IN_ERR] = er; s[ST_IN_SUM] = sm; s[ST_IN_SSQ] = sq; d.st_max[(size_t)v * 2 + 1] = mx;
            } else {
                dg = s[ST_OUT_DEG + side]; c = s[ST_OUT_CNT + side]; er = s[ST_OUT_ERR + side]; sm = s[ST_OUT_SUM + side]; sq = s[ST_OUT_SSQ + side];
                mx = d.st_max[(size_t)v * 2 + side];
            }
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
