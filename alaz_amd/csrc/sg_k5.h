// sg_k5.h — K5 edge_score: per-node projections (MFMA) and the per-edge score
// Part of the kernel translation unit: included by sg_kernels.h (which holds the shared helpers), in this order.
#pragma once

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
                if (v0 + row < N) { d.P[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jb + i)] = p[r]; d.Q[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jb + i)] = q[r]; }
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
                for (int c = 0; c < 4; c++) { d.P[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jq + c)] = p[c]; d.Q[(size_t)vid[row] * SG_F_HID + SG_PQ_POS(jq + c)] = q[c]; }
        }
        __syncthreads();
    }
}

// One wave scores 4 edges per step: 16 lanes per edge, lane q of a group owns hidden units q + 16 m (m = 0..3).
//   t_j = P[u][j] + Q[v][j] + sum_k e_k We[k][j] (fmaf chain over k), ReLU, * w2[j]
//   sum over j in the canonical butterfly order (strides 32, 16, 8, 4, 2, 1): strides 32 and 16 pair units of
//   the SAME lane (j ^ 32 <-> m ^ 2, j ^ 16 <-> m ^ 1), strides 8..1 are DPP steps inside the group's row of 16 lanes — the
//   same additions in the same order as one lane per unit (fp32 addition commutes bitwise), at a quarter of the
//   instructions per edge.  Lane 0 of a group writes the edge's row.
static_assert(sizeof(sg_edge_out) == 64 && offsetof(sg_edge_out, sum_ns) == 0 && offsetof(sg_edge_out, max_ns) == 8 && offsetof(sg_edge_out, sumsq_us) == 16 &&
              offsetof(sg_edge_out, from_ref) == 24 && offsetof(sg_edge_out, to_ref) == 28 && offsetof(sg_edge_out, count) == 32 && offsetof(sg_edge_out, err_count) == 36 &&
              offsetof(sg_edge_out, score) == 40 && offsetof(sg_edge_out, lat_z) == 44 && offsetof(sg_edge_out, err_ratio) == 48 && offsetof(sg_edge_out, alive) == 52 &&
              offsetof(sg_edge_out, p50_us) == 56 && offsetof(sg_edge_out, p99_us) == 60, "k5_edge_score writes a row as eight 8-byte words");
template <bool RESET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k5_edge_score(Dev d, const float* __restrict__ Wh) {
    const u32 E = (u32)d.ctr[C_N_EDGES], nk = (u32)d.ctr[C_N_KNOWN], nl = (u32)d.ctr[C_N_LABELS];
    const float* __restrict__ We = Wh + 2 * SG_F_HID * SG_F_HID;
    const float* __restrict__ w2 = We + SG_F_EDGE * SG_F_HID + SG_F_HID;
    const float b2 = w2[SG_F_HID];
    const u32 lane = threadIdx.x & 63, q = lane & 15, g = lane >> 4;
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    float we[SG_F_EDGE][4], w2r[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        w2r[m] = w2[q + 16 * m];
#pragma unroll
        for (int k = 0; k < (int)SG_F_EDGE; k++) we[k][m] = We[k * SG_F_HID + q + 16 * m];
    }
    // Two steps of four edges per iteration, and the endpoints of the NEXT iteration's edges are fetched while this one's
    // rows are gathered: the dependent chain per iteration is one round trip (the gathers), not two (ids, then gathers).
    // What the 16 lanes of a group need in common is loaded ONCE per group and spread with DPP row_newbcast (a v_mov per value):
    // the two edges' feature vectors are one dword per lane (lanes 0..7 edge A's e_0..e_7, lanes 8..15 edge B's) instead of four
    // 16-byte loads that return the same 32 bytes to all sixteen lanes, the four endpoint ids one dword in lanes 0..3 instead of
    // four loads — 9 memory instructions and 5.5 KiB returned per wave and iteration instead of 15 and 10 KiB (the kernel is
    // bound by the vector-memory pipe, not by HBM: P and Q are L2-resident).
    auto step = [&](const float4 P4, const float4 Q4, const float (&ek)[SG_F_EDGE]) -> float {
        const float pq[4] = {P4.x + Q4.x, P4.y + Q4.y, P4.z + Q4.z, P4.w + Q4.w};
        float r[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            float x = pq[m];
#pragma unroll
            for (int k = 0; k < (int)SG_F_EDGE; k++) x = fmaf(ek[k], we[k][m], x);
            x = x > 0.0f ? x : 0.0f;
            r[m] = x * w2r[m];
        }
        float sum = (r[0] + r[2]) + (r[1] + r[3]);                  // strides 32, then 16
        sum = sum + xor_partner_f32(sum, 8); sum = sum + xor_partner_f32(sum, 4);
        sum = sum + xor_partner_f32(sum, 2); sum = sum + xor_partner_f32(sum, 1);
        return sum;                                                  // (every lane of the 16 holds it)
    };
    static_assert(SG_F_EDGE == 8, "k5_edge_score spreads two 8-float edge feature vectors over a DPP row of 16 lanes");
    const u32* __restrict__ idsrc = (q & 1u) ? d.col : d.csr_from;   // lane q & 3 of a group: from(A), to(A), from(B), to(B)
    // The ROWS of an iteration's eight edges are written by the whole wave: 8 x 64 bytes = 64 lanes x 8 bytes, lane l holds
    // 8-byte word l % 8 of edge l / 8 — one fully coalesced store per iteration instead of four 16-byte stores from one lane in
    // sixteen per step (whose ~100 instructions of row assembly ran with 4 of 64 lanes active).
    //   word 0..2 sum_ns, max_ns, sumsq_us = accumulators 1..3; word 3 from_ref | to_ref; word 4 count | err = accumulator 0;
    //   word 5 score | lat_z; word 6 err_ratio | alive; word 7 p50_us | p99_us
    const u32 wk = lane & 7u, we8 = lane >> 3;
    const u32* __restrict__ srcA = wk == 3 ? d.csr_from : reinterpret_cast<const u32*>(d.errr);
    const u32* __restrict__ srcB = wk == 3 ? d.col : (wk == 5 ? reinterpret_cast<const u32*>(d.latz) : d.alive_csr);
    const u32 accj = wk < 3 ? wk + 1 : 0u;
    const int srcl = (int)(((we8 & 3u) << 4) | (we8 < 4 ? 0u : 8u));     // lane that holds the score sum of this lane's edge: group (edge & 3), its
                                                                         // lower half for the first step's four edges, its upper half for the second's
    if (E) {
        const u32 stride = nw * 8, last = E - 1;
        u32 pa = wave * 8 + g, pb = pa + 4;                          // this iteration's two edges of the lane group (clamped when beyond E)
        // Round 4: TWO iterations in flight.  A wave runs ~30 iterations at C3 and an iteration was one exposed round trip (the gathers:
        // ~2 us) beside ~0.4 us of arithmetic — 73 us of which 60 were latency at four waves per SIMD.  Now the gathers, the feature
        // dword and the row words of iteration i + 1 are issued BEFORE iteration i is computed (a second register set: 21 dwords), its
        // endpoint ids having been fetched an iteration earlier still; loads return in order, so waiting for set i does not wait for
        // set i + 1.  (The set beyond the last iteration is loaded from clamped addresses and dropped.)
        struct K5Set { float4 PA, QA, PB, QB; u32 ew, wa, wb; u64 wacc; };
        auto ids_of = [&](u32 xa, u32 xb) -> u32 { const u32 px = (q & 2u) ? xb : xa; return idsrc[px < E ? px : last]; };
        auto issue = [&](u32 idw_, u32 xa, u32 xb, u32 x0, K5Set& S) {
            u32 ua = dpp32b<0x150>(idw_), va = dpp32b<0x151>(idw_), ub = dpp32b<0x152>(idw_), vb = dpp32b<0x153>(idw_);
            if SG_ABL(d, 0x1000u) { ua &= 15u; va &= 15u; ub &= 15u; vb &= 15u; }   // (diagnostic: gathers that hit the L1)
            const u32 ca = xa < E ? xa : last, cb = xb < E ? xb : last;
            S.PA = reinterpret_cast<const float4*>(d.P + (size_t)ua * SG_F_HID)[q]; S.QA = reinterpret_cast<const float4*>(d.Q + (size_t)va * SG_F_HID)[q];
            S.PB = reinterpret_cast<const float4*>(d.P + (size_t)ub * SG_F_HID)[q]; S.QB = reinterpret_cast<const float4*>(d.Q + (size_t)vb * SG_F_HID)[q];
            S.ew = __float_as_uint(d.efeat[(size_t)(q < 8 ? ca : cb) * SG_F_EDGE + (q & 7u)]);
            // what this lane's row word is made of: fetched beside the gathers
            const u32 er_ = x0 + we8, ec_ = er_ < E ? er_ : last;
            S.wacc = d.acc_csr[(size_t)ec_ * 4 + accj];
            S.wa = srcA[ec_]; S.wb = srcB[ec_];
        };
        // One iteration: `cur` is computed and stored, `nxt` issued; the two sets ALTERNATE between the two calls of the loop body — rotating
        // them through copies at the back edge made the compiler wait for the set in flight there (a v_mov needs its source loaded).
        auto iter = [&](K5Set& cur, K5Set& nxt, const u32 idw_next, u32& idw_after, const u32 p0) {
            const u32 na = pa + stride, nb = pb + stride;
            // the endpoints of the iteration after the next go out FIRST: they are then older than the gathers issued below, and the next
            // iteration's wait for them does not wait for those gathers (issued the other way round, it was a vmcnt(0) at the loop top)
            idw_after = ids_of(na + stride, nb + stride);
            issue(idw_next, na, nb, p0 + stride, nxt);
            __builtin_amdgcn_sched_barrier(0);                       // (the scheduler moved the arithmetic of `cur` above these loads)
            const float4 PA = cur.PA, QA = cur.QA, PB = cur.PB, QB = cur.QB;
            const u32 ew = cur.ew, wa = cur.wa, wb = cur.wb; const u64 wacc = cur.wacc;
            const u32 er = p0 + we8, ec = er < E ? er : last;
            const float eka[SG_F_EDGE] = {__uint_as_float(dpp32b<0x150>(ew)), __uint_as_float(dpp32b<0x151>(ew)), __uint_as_float(dpp32b<0x152>(ew)), __uint_as_float(dpp32b<0x153>(ew)),
                                          __uint_as_float(dpp32b<0x154>(ew)), __uint_as_float(dpp32b<0x155>(ew)), __uint_as_float(dpp32b<0x156>(ew)), __uint_as_float(dpp32b<0x157>(ew))};
            const float ekb[SG_F_EDGE] = {__uint_as_float(dpp32b<0x158>(ew)), __uint_as_float(dpp32b<0x159>(ew)), __uint_as_float(dpp32b<0x15A>(ew)), __uint_as_float(dpp32b<0x15B>(ew)),
                                          __uint_as_float(dpp32b<0x15C>(ew)), __uint_as_float(dpp32b<0x15D>(ew)), __uint_as_float(dpp32b<0x15E>(ew)), __uint_as_float(dpp32b<0x15F>(ew))};
            const float sa = step(PA, QA, eka);
            const float sb = step(PB, QB, ekb);
            const float mysum = __shfl(q < 8 ? sa : sb, srcl, 64);     // ONE shuffle executed by all lanes (two under a select were sunk into exec-masked
                                                                       // branches by the compiler: ds_bpermute returns 0 for an inactive source lane)
            u64 val = wacc;                                          // words 0..2 and 4
            if (wk == 3) val = (u64)ref_of_dense(wa, nk, nl) | ((u64)ref_of_dense(wb, nk, nl) << 32);
            else if (wk == 5) { const float logit = mysum + b2; val = (u64)__float_as_uint(1.0f / (1.0f + expf(-logit))) | ((u64)wb << 32); }
            else if (wk == 6) val = (u64)wa | ((u64)wb << 32);
            else if (wk == 7) {
                val = 0;
                if (d.hist) {                                        // percentiles off the log2 histogram (include/servicegraph.h)
                    const ulonglong2* __restrict__ ac = reinterpret_cast<const ulonglong2*>(d.acc_csr + (size_t)ec * 4);
                    const u32 count = (u32)(ac[0].x & 0xFFFFFFFFull); const u64 max_ns = ac[1].x;
                    if (count) {
                        const uint4* hp = reinterpret_cast<const uint4*>(d.hist_csr + (size_t)ec * SG_HIST_BINS);
                        const uint4 h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3];
                        const u32 hb[SG_HIST_BINS] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
                        u64 r50 = ((u64)count * 50 + 99) / 100, r99 = ((u64)count * 99 + 99) / 100;
                        r50 = r50 ? r50 : 1; r99 = r99 ? r99 : 1;
                        u64 cum = 0; u32 b50 = SG_HIST_BINS - 1, b99 = SG_HIST_BINS - 1; bool f50 = false, f99 = false;
#pragma unroll
                        for (u32 b = 0; b < SG_HIST_BINS; b++) { cum += hb[b]; if (!f50 && cum >= r50) { b50 = b; f50 = true; } if (!f99 && cum >= r99) { b99 = b; f99 = true; } }
                        u64 e50 = b50 == SG_HIST_BINS - 1 ? max_ns : (1ull << (17 + b50)), e99 = b99 == SG_HIST_BINS - 1 ? max_ns : (1ull << (17 + b99));
                        e50 = e50 > max_ns ? max_ns : e50; e99 = e99 > max_ns ? max_ns : e99;
                        e50 /= 1000ull; e99 /= 1000ull;
                        val = (u64)(e50 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)e50) | ((u64)(e99 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)e99) << 32);
                    }
                }
            }
            // The store as a BUFFER store on the wave's 512 bytes of this iteration (p0 is wave-uniform): rows beyond E are dropped by
            // the resource's range check, not by a branch — a branch around the store is a join for the compiler's vmcnt bookkeeping,
            // and the next half-iteration's wait for its endpoint ids then also waited for the first gather of the set in flight.
            {
                const u32 p0u = (u32)__builtin_amdgcn_readfirstlane((int)p0);
                const u32 nrow = SG_ABL(d, 0x2000u) ? 0u : (E - p0u < 8u ? E - p0u : 8u);
                const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u64*>(d.rows) + (size_t)p0u * 8, 0, (int)(nrow * 64u), 0x00020000);
                v2u_t dv; dv.x = (u32)val; dv.y = (u32)(val >> 32);
                __builtin_amdgcn_raw_buffer_store_b64(dv, rr_, lane * 8u, 0, 0);
            }
            pa = na; pb = nb;
        };
        K5Set A, B;
        u32 i1 = ids_of(pa + stride, pb + stride), i2;               // the endpoints of iteration 1 ...
        { const u32 id0 = ids_of(pa, pb); issue(id0, pa, pb, wave * 8, A); }   // ... in flight before iteration 0's gathers
        for (u32 p0 = wave * 8; p0 < E; p0 += 2 * stride) {
            iter(A, B, i1, i2, p0);
            if (p0 + stride >= E) break;                             // (uniform)
            iter(B, A, i2, i1, p0 + stride);
        }
    }
    if (RESET) {
        // Window reset folded into the last kernel of the pipeline (nothing after it reads these arrays;
        // the counters stay: sg_window_read / the next kc_prepare consume them).
        const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, nt = (u64)gridDim.x * 256;
        const u64 nc = (u64)d.ncap + 1;
        if (!d.dh_g) for (u64 i = tid; i < nc * SG_DEG_REP; i += nt) { d.deg[i * SG_DEG_STRIDE] = 0; if (d.warm) d.deg2[i * SG_DEG_STRIDE] = 0; }
        for (u64 i = tid; i < nc; i += nt) d.cursor[i] = 0;
        for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_SUM_WORDS; i += nt) d.st_sum[i] = 0;
        for (u64 i = tid; i < (u64)d.ncap * SG_NODE_STAT_MAX_WORDS; i += nt) d.st_max[i] = 0;
        for (u64 i = tid; i <= d.obmask; i += nt) d.obkeys[i] = 0;
    }
}
