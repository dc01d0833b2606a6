// store_probe.hip — micro-measurements behind the K1 round-2 design (not part of the product): how fast can 256 workgroups
// append 16-byte records to many per-(partition, workgroup) pieces, as a function of how the stores are shaped, and how
// fast are LDS u64 / u32 atomics on (nearly) distinct addresses.  Build: hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ u32 mix(u32 h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

// MODE 0: one 16-B store per lane to piece (p, w), p random in [0, np): the K1a pattern (position from an LDS counter)
// MODE 1: same, store with nt            MODE 2: same, store with sc0 sc1 (write-through, no L2 line kept)
// MODE 3: groups of 4 lanes write 4 consecutive 16-B slots of ONE piece (64 B contiguous per group, one instruction)
// MODE 4: ONE lane in 4 writes 64 B as 4 consecutive dwordx4 instructions (same bytes as mode 3, 4 instructions, 16 lanes active)
// MODE 5: groups of 8 lanes write 128 B contiguous
template <int MODE>
__global__ __launch_bounds__(1024) void k_store(uint4* slab, u32 np, u32 pslots, u32 iters) {
    extern __shared__ u32 cnt[];
    const u32 w = blockIdx.x, t = threadIdx.x;
    for (u32 p = t; p < np; p += 1024) cnt[p] = 0;
    __syncthreads();
    u32 seed = mix(w * 1024 + t + 1);
    for (u32 it = 0; it < iters; it++) {
        seed = mix(seed + it);
        const uint4 rec = make_uint4(seed, it, w, t);
        if (MODE <= 2) {
            const u32 p = seed & (np - 1);
            const u32 pos = atomicAdd(&cnt[p], 1u) % pslots;
            uint4* dst = slab + ((size_t)p * gridDim.x + w) * pslots + pos;
            if (MODE == 0) *dst = rec;
            else { typedef u32 v4 __attribute__((ext_vector_type(4))); const v4 r = {rec.x, rec.y, rec.z, rec.w};
                if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(r) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(r) : "memory"); }
        } else if (MODE == 3 || MODE == 5) {
            const u32 G = MODE == 3 ? 4 : 8;
            const u32 leader = __shfl(seed, (t & 63) & ~(G - 1), 64);
            const u32 p = leader & (np - 1);
            u32 pos = 0;
            if ((t & (G - 1)) == 0) pos = atomicAdd(&cnt[p], G) % pslots;
            pos = __shfl(pos, (t & 63) & ~(G - 1), 64) + (t & (G - 1));
            uint4* dst = slab + ((size_t)p * gridDim.x + w) * pslots + (pos % pslots);
            *dst = rec;
        } else if (MODE == 6) {                                      // one record per lane in a 32-byte slot: record + 16 bytes of padding, two stores
            const u32 p = seed & (np - 1);
            const u32 pos = atomicAdd(&cnt[p], 1u) % (pslots / 2);
            uint4* dst = slab + ((size_t)p * gridDim.x + w) * pslots + 2 * pos;
            dst[0] = rec; dst[1] = make_uint4(0, 0, 0, 0);
        } else if (MODE == 7) {                                      // one record per LANE PAIR in a 32-byte slot: one instruction writes the whole sector
            const u32 leader = __shfl(seed, (t & 63) & ~1u, 64);
            const u32 p = leader & (np - 1);
            u32 pos = 0;
            if ((t & 1) == 0) pos = atomicAdd(&cnt[p], 1u) % (pslots / 2);
            pos = __shfl(pos, (t & 63) & ~1u, 64);
            uint4* dst = slab + ((size_t)p * gridDim.x + w) * pslots + 2 * pos + (t & 1);
            *dst = rec;
        } else {
            if ((t & 3) == 0) {
                const u32 p = seed & (np - 1);
                const u32 pos = atomicAdd(&cnt[p], 4u) % pslots;
                uint4* dst = slab + ((size_t)p * gridDim.x + w) * pslots + pos;
                dst[0] = rec; dst[1] = rec; dst[2] = rec; dst[3] = rec;
            }
        }
    }
}

// LDS atomics: each lane adds to slot (hash & mask): WIDE = 4 x u64 per item (K1 accumulators), else u32 count + 2 x u64
template <int KIND>
__global__ __launch_bounds__(1024) void k_lds(u32 mask, u32 iters, u64* out) {
    extern __shared__ u64 tab[];
    for (u32 i = threadIdx.x; i < (mask + 1) * 4; i += 1024) tab[i] = 0;
    __syncthreads();
    u32 seed = mix(blockIdx.x * 1024 + threadIdx.x + 1);
    for (u32 it = 0; it < iters; it++) {
        seed = mix(seed + it);
        u64* a = tab + (size_t)(seed & mask) * 4;
        if (KIND == 0) { atomicAdd(&a[0], 1ull); atomicAdd(&a[1], (u64)seed); atomicMax(&a[2], (u64)seed); atomicAdd(&a[3], (u64)seed * seed); }
        else if (KIND == 1) { atomicAdd((u32*)&a[0], 1u); atomicAdd(&a[1], (u64)seed); atomicAdd(&a[3], (u64)seed * seed); if (__hip_atomic_load(&a[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < seed) atomicMax(&a[2], (u64)seed); }
        else if (KIND == 2) { atomicAdd((u32*)&a[0], 1u); }
        else { atomicAdd(&a[1], (u64)seed); }
    }
    __syncthreads();
    if (threadIdx.x == 0 && tab[5] == 0x1234567) *out = tab[5];
}

static double run(const char* name, int reps, std::function<void()> f, double items) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double us = ms * 1000.0 / reps;
    printf("%-64s %9.2f us  %8.2f G items/s\n", name, us, items / us / 1e3);
    return us;
}
int main() {
    const u32 nwg = 256, pslots = 128, iters = 32;           // 256 x 1024 x 32 = 8.4 M records per launch (C3 scale)
    uint4* slab; CK(hipMalloc(&slab, (size_t)4096 * nwg * pslots * 16));
    u64* out; CK(hipMalloc(&out, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const double items = (double)nwg * 1024 * iters;
    for (u32 np : {64u, 256u, 1024u, 2048u, 4096u}) {
        char nm[128];
#define RUN(M, txt) snprintf(nm, sizeof nm, "np=%-4u " txt, np); run(nm, 5, [&] { hipLaunchKernelGGL(k_store<M>, dim3(nwg), dim3(1024), np * 4, 0, slab, np, pslots, iters); }, items)
        RUN(0, "16 B per lane, random piece (plain)");
        RUN(1, "16 B per lane, random piece (nt)");
        RUN(2, "16 B per lane, random piece (sc0 sc1)");
        RUN(3, "4 lanes x 16 B contiguous, one instruction");
        RUN(4, "1 lane x 4 dwordx4 contiguous (64 B), 16 lanes active");
        RUN(5, "8 lanes x 16 B contiguous, one instruction");
        RUN(6, "16 B record + 16 B pad per lane (32-B slot, two stores)");
        RUN(7, "2 lanes x 16 B (32-B slot, one instruction; half the records)");
    }
    for (u32 slots : {1024u, 2048u, 4096u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "LDS %u slots: 4 x u64 atomics per item", slots); run(nm, 5, [&] { hipLaunchKernelGGL(k_lds<0>, dim3(nwg), dim3(1024), slots * 32, 0, slots - 1, 64u, out); }, (double)nwg * 1024 * 64);
        snprintf(nm, sizeof nm, "LDS %u slots: u32 + 2 x u64 + conditional max", slots); run(nm, 5, [&] { hipLaunchKernelGGL(k_lds<1>, dim3(nwg), dim3(1024), slots * 32, 0, slots - 1, 64u, out); }, (double)nwg * 1024 * 64);
        snprintf(nm, sizeof nm, "LDS %u slots: one u32 atomic per item", slots); run(nm, 5, [&] { hipLaunchKernelGGL(k_lds<2>, dim3(nwg), dim3(1024), slots * 32, 0, slots - 1, 64u, out); }, (double)nwg * 1024 * 64);
        snprintf(nm, sizeof nm, "LDS %u slots: one u64 atomic per item", slots); run(nm, 5, [&] { hipLaunchKernelGGL(k_lds<3>, dim3(nwg), dim3(1024), slots * 32, 0, slots - 1, 64u, out); }, (double)nwg * 1024 * 64);
    }
    return 0;
}
