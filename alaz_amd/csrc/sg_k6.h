// sg_k6.h — K6 halo: request lists, active lists, pack / unpack (multi-GPU)
// Part of the kernel translation unit: included by sg_kernels.h (which holds the shared helpers), in this order.
#pragma once

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
