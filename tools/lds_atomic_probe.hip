// lds_atomic_probe.hip — what an LDS atomic costs on gfx950 (not part of the product): lane-operations per clock and CU for the forms pass B
// of K1 and K3's scan use, against plain LDS reads / writes.  Sizes the merge of k1b_stream_merge.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip      Run: ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

constexpr int HT = 2048;           // table slots, as pass B's sub-table
enum Op { ADD32, ADD64, MAX32, ADD32_RTN, READ32, WRITE32, ADD32_QUARTER, ADD32_LINEAR, ADD64_LINEAR, ADD32_SAME, MIX_PASSB, MIX_2OPS, READ64, ADD64_QUARTER };

__device__ __forceinline__ u32 mixh(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int OP>
__global__ __launch_bounds__(1024) void k_probe(u32 iters, u64* sink) {
    __shared__ u64 acc[4 * HT];        // [4][HT] as pass B
    __shared__ u32 key[HT];
    const u32 t = threadIdx.x, lane = t & 63u;
    for (u32 i = t; i < 4 * HT; i += blockDim.x) acc[i] = 0;
    for (u32 i = t; i < HT; i += blockDim.x) key[i] = i;
    __syncthreads();
    u32* acc32 = reinterpret_cast<u32*>(acc);
    u32 s = 0;
    u32 x = mixh(t * 2654435761u + blockIdx.x);
    for (u32 i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const u32 h = (x >> 8) & (HT - 1);
        if constexpr (OP == ADD32) atomicAdd(&acc32[2 * h], 1u);
        if constexpr (OP == ADD64) atomicAdd(&acc[h], (u64)x);
        if constexpr (OP == MAX32) atomicMax(&acc32[2 * h], x);
        if constexpr (OP == ADD32_RTN) s += atomicAdd(&acc32[2 * h], 1u);
        if constexpr (OP == READ32) s += __hip_atomic_load(&key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (OP == READ64) s += (u32)__hip_atomic_load(&acc[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (OP == WRITE32) __hip_atomic_store(&key[h], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (OP == ADD32_QUARTER) { if ((lane & 3u) == (i & 3u)) atomicAdd(&acc32[2 * h], 1u); }
        if constexpr (OP == ADD64_QUARTER) { if ((lane & 3u) == (i & 3u)) atomicAdd(&acc[h], (u64)x); }
        if constexpr (OP == ADD32_LINEAR) atomicAdd(&key[(t + 64u * i) & (HT - 1)], 1u);                   // lane l -> bank l % 32: no conflict beyond the two halves
        if constexpr (OP == ADD64_LINEAR) atomicAdd(&acc[(t + 64u * i) & (HT - 1)], (u64)x);
        if constexpr (OP == ADD32_SAME) atomicAdd(&key[i & (HT - 1)], 1u);                                  // every lane of the workgroup: one address
        if constexpr (OP == MIX_PASSB) {                                                                   // a narrow record of pass B (PACK form)
            const u32 k = __hip_atomic_load(&key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const u32 hh = k & (HT - 1);
            atomicAdd(&acc[HT + hh], (u64)x | (1ull << 48)); atomicMax(&acc32[2 * (2 * HT + hh)], x); atomicAdd(&acc[3 * HT + hh], (u64)(x >> 10) * (x >> 10));
        }
        if constexpr (OP == MIX_2OPS) {
            const u32 k = __hip_atomic_load(&key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const u32 hh = k & (HT - 1);
            atomicAdd(&acc[HT + hh], (u64)x | (1ull << 48)); atomicAdd(&acc[3 * HT + hh], (u64)(x >> 10) * (x >> 10));
        }
    }
    __syncthreads();
    if (s == 0x12345678u || acc[t] == 0x1234567ull) sink[0] = s + acc[t];
}

template <int OP> static void run(const char* name, double ops_per_iter, int wg_per_cu, double mhz) {
    u64* sink; CK(hipMalloc(&sink, 64));
    const u32 iters = 4000; const int grid = 256 * wg_per_cu, nt = 1024;   // one or two 1024-thread workgroups per CU (72 KB of LDS each, as pass B)
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<OP>, dim3(grid), dim3(nt), 0, 0, 100u, sink); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k_probe<OP>, dim3(grid), dim3(nt), 0, 0, iters, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double lane_ops = (double)grid * nt * iters * ops_per_iter, per_cu_ns = lane_ops / 256.0 / (ms * 1e6);
    printf("%-46s %d x %4d thr/CU  %8.1f us   %6.2f lane-ops/ns/CU  = %5.2f per clock at %.0f MHz\n", name, wg_per_cu, nt, ms * 1e3, per_cu_ns, per_cu_ns / (mhz * 1e-3), mhz);
    CK(hipFree(sink));
}

int main(int argc, char** argv) {
    const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
    for (int w : {1, 2}) {
        run<READ32>("ds_read_b32, random slot", 1, w, mhz);
        run<READ64>("ds_read_b64, random slot", 1, w, mhz);
        run<WRITE32>("ds_write_b32, random slot", 1, w, mhz);
        run<ADD32>("ds_add_u32, random slot", 1, w, mhz);
        run<ADD64>("ds_add_u64, random slot", 1, w, mhz);
        run<MAX32>("ds_max_u32, random slot", 1, w, mhz);
        run<ADD32_RTN>("ds_add_rtn_u32, random slot", 1, w, mhz);
        run<ADD32_QUARTER>("ds_add_u32, random, a quarter of the lanes", 0.25, w, mhz);
        run<ADD64_QUARTER>("ds_add_u64, random, a quarter of the lanes", 0.25, w, mhz);
        run<ADD32_LINEAR>("ds_add_u32, lane l -> word l (no conflict)", 1, w, mhz);
        run<ADD64_LINEAR>("ds_add_u64, lane l -> word l", 1, w, mhz);
        run<ADD32_SAME>("ds_add_u32, one address for all lanes", 1, w, mhz);
        run<MIX_PASSB>("pass B record: read + add64 + max32 + add64", 4, w, mhz);
        run<MIX_2OPS>("two-op record: read + add64 + add64", 3, w, mhz);
    }
    return 0;
}
