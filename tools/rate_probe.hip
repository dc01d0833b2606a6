// rate_probe.hip — issue-rate micro-measurements that size the round-4 K1 rewrite on MI355X (not part of the product):
//   * integer VALU: cycles per wave64 instruction and SIMD for the instructions pass A is made of, at 1 / 2 / 4 waves per SIMD
//   * LDS: cycles per wave64 instruction and CU for the accesses pass A / pass B make (random addresses, as in the kernels)
//   * the shader clock under that load (s_memtime ticks per 100 MHz s_memrealtime tick)
// Build: hipcc --offload-arch=gfx950 -O3 -o rate_probe rate_probe.hip        Run on the GPU box: tools/rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

enum { V_ADD, V_XOR, V_AND_OR, V_LSHL_OR, V_MUL24, V_MULLO, V_MULHI, V_CNDMASK, V_CMP_CND, V_ADD64, V_CMP64, V_BFE, V_LSHR64, V_READLANE, V_NOPS };
static const char* vname[] = {"v_add_u32", "v_xor_b32", "v_and_or_b32", "v_lshl_or_b32", "v_mul_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_cndmask_b32 (vcc fixed)",
                              "v_cmp_lt_u32 + v_cndmask", "v_lshl_add_u64", "v_cmp_lt_u64 + 2 v_cndmask", "v_bfe_u32", "v_lshrrev_b64", "v_readlane_b32 (-> s)", "s_nop 0"};

// 8 independent chains, ITER trips of 8 x 4 = 32 instructions
template <int OP>
__global__ void k_valu(u64* out, u32 iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    u32 b = seed | 1u, c = seed * 7 + 3;
    u64 q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    const u64 t0 = __builtin_readcyclecounter();
    for (u32 i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (OP == V_ADD) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == V_XOR) asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == V_AND_OR) asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (OP == V_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %8\n v_lshl_or_b32 %1, %1, 3, %8\n v_lshl_or_b32 %2, %2, 3, %8\n v_lshl_or_b32 %3, %3, 3, %8\n v_lshl_or_b32 %4, %4, 3, %8\n v_lshl_or_b32 %5, %5, 3, %8\n v_lshl_or_b32 %6, %6, 3, %8\n v_lshl_or_b32 %7, %7, 3, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == V_MUL24) asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == V_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == V_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == V_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (OP == V_CMP_CND) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
            if (OP == V_ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0 | 1));
            if (OP == V_CMP64) asm volatile("v_cmp_lt_u64 vcc, %4, %5\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cmp_lt_u64 vcc, %5, %4\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %6, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q0), "v"(q2), "v"(b) : "vcc");
            if (OP == V_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 20\n v_bfe_u32 %1, %1, 3, 20\n v_bfe_u32 %2, %2, 3, 20\n v_bfe_u32 %3, %3, 3, 20\n v_bfe_u32 %4, %4, 3, 20\n v_bfe_u32 %5, %5, 3, 20\n v_bfe_u32 %6, %6, 3, 20\n v_bfe_u32 %7, %7, 3, 20" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (OP == V_LSHR64) asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            if (OP == V_READLANE) asm volatile("v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %1, 2\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 4\n v_readlane_b32 s20, %4, 1\n v_readlane_b32 s21, %5, 2\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 4" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23");
            if (OP == V_NOPS) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (u32)q0 ^ (u32)q1 ^ (u32)q2 ^ (u32)q3) == 0x12345678u) out[1] = a0;
}

// instructions per trip for the rate computation (per chain set)
static int v_per_trip(int op) { return (op == V_CMP_CND) ? 4 * 8 : (op == V_CMP64) ? 4 * 6 : 4 * 8; }

enum { L_RD64, L_RD16, L_RD128, L_WR64, L_WR32, L_ADD32, L_ADD32_RTN, L_ADD32_RTN_512, L_ADD64, L_MAX64, L_MAX32, L_CAS32, L_RD32_DEP };
static const char* lname[] = {"ds_read_b64 random / 8 KiB", "ds_read_u16 random / 32 KiB", "ds_read_b128 random / 40 KiB", "ds_write_b64 random / 64 KiB", "ds_write_b32 random / 32 KiB",
                              "ds_add_u32 (no rtn) random / 2048 ctr", "ds_add_rtn_u32 random / 2048 ctr", "ds_add_rtn_u32 random / 512 ctr", "ds_add_u64 (no rtn) random / 8192 slots",
                              "ds_max_u64 (no rtn) random / 8192 slots", "ds_max_u32 (no rtn) random / 8192", "ds_cmpst_rtn_b32 random / 2048", "ds_read_b32 DEPENDENT chain (latency)"};
template <int OP>
__global__ void k_lds(u64* out, u32 iters, u32 seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    for (u32 i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<u32*>(sm)[i] = (i * 2654435761u) >> 8;
    __syncthreads();
    u32 x = (threadIdx.x * 2654435761u) ^ seed, acc = 0;
    u64 acc64 = 0;
    const u64 t0 = __builtin_readcyclecounter();
    for (u32 i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            x = x * 1664525u + 1013904223u;
            const u32 h = x >> 8;
            if (OP == L_RD64) acc64 += reinterpret_cast<const u64*>(sm)[h & 1023u];
            if (OP == L_RD16) acc += reinterpret_cast<const unsigned short*>(sm)[h & 16383u];
            if (OP == L_RD128) { const uint4 v = reinterpret_cast<const uint4*>(sm)[h % 2560u]; acc += v.x ^ v.w; }
            if (OP == L_WR64) reinterpret_cast<u64*>(sm)[h & 8191u] = (u64)x | ((u64)i << 32);
            if (OP == L_WR32) reinterpret_cast<u32*>(sm)[h & 8191u] = x;
            if (OP == L_ADD32) atomicAdd(&reinterpret_cast<u32*>(sm)[h & 2047u], 1u);
            if (OP == L_ADD32_RTN) acc += atomicAdd(&reinterpret_cast<u32*>(sm)[h & 2047u], 1u);
            if (OP == L_ADD32_RTN_512) acc += atomicAdd(&reinterpret_cast<u32*>(sm)[h & 511u], 1u);
            if (OP == L_ADD64) atomicAdd(&reinterpret_cast<u64*>(sm)[h & 8191u], (u64)x);
            if (OP == L_MAX64) atomicMax(&reinterpret_cast<u64*>(sm)[h & 8191u], (u64)x);
            if (OP == L_MAX32) atomicMax(&reinterpret_cast<u32*>(sm)[h & 8191u], x);
            if (OP == L_CAS32) acc += atomicCAS(&reinterpret_cast<u32*>(sm)[h & 2047u], 0xFFFFFFFFu, x);
            if (OP == L_RD32_DEP) { acc = reinterpret_cast<const u32*>(sm)[(acc + threadIdx.x) & 16383u]; }
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if ((acc ^ (u32)acc64) == 0x12345678u) out[1] = acc;
}

// the shader clock: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), every CU busy with VALU work meanwhile
__global__ void k_clock(u64* out, u32 iters) {
    u32 a = threadIdx.x, b = a * 3 + 1, c = a * 5 + 7;
    const u64 c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (u32 i = 0; i < iters; i++) { a = a * 1664525u + b; b = b * 22695477u + c; c ^= a >> 3; }
    const u64 c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if ((a ^ b ^ c) == 0x12345678u) out[2] = a;
}

template <typename F> static double timed_us(F f, int reps = 3) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1000.0 / reps;
}

template <int OP> static void valu_case(u64* d_out) {
    const u32 iters = 2000;
    printf("%-30s", vname[OP]);
    for (int wps : {1, 2, 4}) {                                  // waves per SIMD: 256 / 512 / 1024-thread workgroups, one per CU
        timed_us([&] { hipLaunchKernelGGL((k_valu<OP>), dim3(256), dim3(256 * wps), 0, 0, d_out, iters, 12345u); }, 2);
        u64 h[2]; CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
        const double instr = (double)iters * v_per_trip(OP) * wps;   // wave-instructions issued on one SIMD
        printf("  %dw/SIMD %6.2f cyc/instr", wps, (double)h[0] / instr);
    }
    printf("\n");
}
template <int OP> static void lds_case(u64* d_out) {
    const u32 iters = 400;
    printf("%-44s", lname[OP]);
    for (int waves : {4, 16}) {                                  // waves per CU (one workgroup per CU)
        timed_us([&] { hipLaunchKernelGGL((k_lds<OP>), dim3(256), dim3(64 * waves), 65536, 0, d_out, iters, 777u); }, 2);
        u64 h[2]; CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
        const double instr = (double)iters * 8 * waves;          // wave-instructions the CU's LDS served
        printf("  %2d waves/CU %7.2f cyc/wave-instr", waves, (double)h[0] / instr);
    }
    printf("\n");
}

int main() {
    u64* d_out; CK(hipMalloc(&d_out, 64)); CK(hipMemset(d_out, 0, 64));
    for (int rep = 0; rep < 3; rep++) {
        const double us = timed_us([&] { hipLaunchKernelGGL(k_clock, dim3(1024), dim3(256), 0, 0, d_out, 200000u); }, 1);
        u64 h[2]; CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
        printf("clock probe %d: %.0f us, %llu shader cycles / %llu realtime ticks -> effective sclk %.0f MHz\n", rep, us, h[0], h[1], h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0);
    }
    printf("---- integer VALU, cycles per wave64 instruction on one SIMD (lower = faster; 8 independent chains per wave)\n");
    valu_case<V_ADD>(d_out); valu_case<V_XOR>(d_out); valu_case<V_AND_OR>(d_out); valu_case<V_LSHL_OR>(d_out); valu_case<V_MUL24>(d_out); valu_case<V_MULLO>(d_out);
    valu_case<V_MULHI>(d_out); valu_case<V_CNDMASK>(d_out); valu_case<V_CMP_CND>(d_out); valu_case<V_ADD64>(d_out); valu_case<V_CMP64>(d_out); valu_case<V_BFE>(d_out);
    valu_case<V_LSHR64>(d_out); valu_case<V_READLANE>(d_out); valu_case<V_NOPS>(d_out);
    printf("---- LDS, cycles per wave64 instruction and CU (includes the address arithmetic of the probe loop: ~6 VALU per access)\n");
    lds_case<L_RD64>(d_out); lds_case<L_RD16>(d_out); lds_case<L_RD128>(d_out); lds_case<L_WR64>(d_out); lds_case<L_WR32>(d_out); lds_case<L_ADD32>(d_out); lds_case<L_ADD32_RTN>(d_out);
    lds_case<L_ADD32_RTN_512>(d_out); lds_case<L_ADD64>(d_out); lds_case<L_MAX64>(d_out); lds_case<L_MAX32>(d_out); lds_case<L_CAS32>(d_out); lds_case<L_RD32_DEP>(d_out);
    return 0;
}
