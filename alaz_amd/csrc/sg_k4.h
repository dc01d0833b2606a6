// sg_k4.h — K4 sage_layer: neighbour gather-mean and the dense tiles (MFMA)
// Part of the kernel translation unit: included by sg_kernels.h (which holds the shared helpers), in this order.
#pragma once

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

// The same tile with the B operands fetched up front: dense_tile_load_b issues the lane's K / 4 loads of W together (ONE round trip, and
// none of it depends on the tile's rows, so k4_sage_layer issues them before its phase 1), dense_tile_mfma_pre runs the chain — the same
// MFMAs in the same k order.  (dense_tile_mfma waits for its loads four at a time: K / 16 dependent round trips per call — eight for a
// 64-feature layer, four more for the projection: most of what the dense launches took.)
template <int K>
__device__ __forceinline__ void dense_tile_load_b(const float* __restrict__ W, int jb, float (&b)[K / 4]) {
    const int l = threadIdx.x & 63, i = l & 15, kq = l >> 4;
#pragma unroll
    for (int kb = 0; kb < K / 4; kb++) b[kb] = W[(size_t)(kb * 4 + kq) * SG_F_HID + jb + i];
}
template <int K>
__device__ __forceinline__ f32x4 dense_tile_mfma_pre(const float* A, int lda, const float (&b)[K / 4], f32x4 c) {
    const int l = threadIdx.x & 63, i = l & 15, kq = l >> 4;
#pragma unroll
    for (int kb = 0; kb < K / 4; kb++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * lda + kb * 4 + kq], b[kb], c, 0, 0, 0);
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
    // (MFMA build: the weights of the wave's 16 columns — and of its share of the projection — are fetched once, ahead of the first tile's rows)
    float bw[USE_MFMA ? 2 * FI / 4 : 1], bpj[USE_MFMA && PROJ ? SG_F_HID / 4 : 1];
    float bj0 = 0.0f, bj1 = 0.0f;
    if constexpr (USE_MFMA) {
        if (wave < 4) { dense_tile_load_b<2 * FI>(Wl, (int)wave * 16, bw); bj0 = bias[wave * 16 + (lane & 15)]; }
        if constexpr (PROJ) if (wave < 8) {
            const float* __restrict__ Wu = Wh; const float* __restrict__ Wv = Wh + SG_F_HID * SG_F_HID;
            dense_tile_load_b<SG_F_HID>(wave >= 4 ? Wv : Wu, (int)(wave & 3) * 16, bpj);
            bj1 = wave >= 4 ? 0.0f : (Wv + SG_F_HID * SG_F_HID + SG_F_EDGE * SG_F_HID)[(wave & 3) * 16 + (lane & 15)];
        }
    }
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
            else if constexpr (PRE) {
                static_assert(!PRE || FI <= 64, "one element of the self row and of the mean per lane");
                // The row's five loads — its self element, its two row pointers, its element of the mean, its first work item — are issued
                // TOGETHER, whatever the degree turns out to be (every address is valid; lanes beyond FI read element 0): one round trip per
                // row where the mean behind the degree and the work item behind the comparison were trips of their own; a hub row's block
                // sums eight at a time (a load behind every addition: nine dependent trips for a row of 3 750 edges, and the tile with such
                // a row was the launch's tail).
                const u32 kl = lane < (u32)FI ? lane : 0u;
                const float xs = hin[(size_t)v * FI + kl];
                const u32 beg = d.rowptr[v], end = d.rowptr[v + 1];
                const float xm = d.nmean[(size_t)v * SG_F_HID + kl];
                const u32 ib = d.hub_base[v];
                const u32 dg = end - beg;
                float mean = dg ? xm : 0.0f;
                if (dg > SG_MEAN_BLOCK) {                            // a hub row: its block sums (k4_gather's work items) in block order, one division
                    const u32 nblk = (dg + SG_MEAN_BLOCK - 1) / SG_MEAN_BLOCK;
                    float total = 0.0f;
                    for (u32 j0 = 0; j0 < nblk; j0 += 8) {
                        float x[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) x[q] = d.hub_part[(size_t)(ib + (j0 + q < nblk ? j0 + q : nblk - 1)) * SG_F_HID + kl];
#pragma unroll
                        for (int q = 0; q < 8; q++) if (j0 + q < nblk) total = (j0 + q) ? total + x[q] : x[q];
                    }
                    mean = total / (float)dg;
                }
                if (lane < (u32)FI) { row[lane] = xs; row[FI + lane] = mean; }
            } else {
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
            if constexpr (USE_MFMA) {
                const int jb = wave * 16, i = lane & 15;
                f32x4 c = { bj0, bj0, bj0, bj0 };
                c = dense_tile_mfma_pre<2 * FI>(A, LDA, bw, c);
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
        if constexpr (PROJ) {
            __syncthreads();
            // waves 0..3 -> P columns, waves 4..7 -> Q columns
            const float* __restrict__ Wu = Wh; const float* __restrict__ Wv = Wh + SG_F_HID * SG_F_HID;
            const float* __restrict__ b1 = Wv + SG_F_HID * SG_F_HID + SG_F_EDGE * SG_F_HID;
            if (wave < 8) {
                const bool isq = wave >= 4;
                const int jb = (wave & 3) * 16, i = lane & 15;
                float* dst = isq ? d.Q : d.P;
                const float* __restrict__ Wm = isq ? Wv : Wu;
                if constexpr (USE_MFMA) {
                    f32x4 c = { bj1, bj1, bj1, bj1 };
                    c = dense_tile_mfma_pre<SG_F_HID>(H, LDH, bpj, c);
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
