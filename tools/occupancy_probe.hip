// occupancy_probe.hip — do two workgroups share a CU?  (measurement behind pass B's design; not part of the product)
// Each workgroup records wall_clock64 at start and end around a fixed-length spin; the host counts, per launch, how many
// workgroups were alive at the median start time.  Build: hipcc --offload-arch=gfx950 -O3 -o occupancy_probe occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)
// SG: the highest SGPR the kernel is made to use (clobber list) — does the scalar register file limit co-residency?
template <int NT, int SG = 0>
__global__ __launch_bounds__(NT) void k_spin(u64* st, u32 ticks) {
    extern __shared__ u32 lds[];
    if (SG == 39) asm volatile("" ::: "s39");
    if (SG == 47) asm volatile("" ::: "s47");
    if (SG == 55) asm volatile("" ::: "s55");
    if (SG == 63) asm volatile("" ::: "s63");
    if (SG == 71) asm volatile("" ::: "s71");
    if (SG == 79) asm volatile("" ::: "s79");
    if (SG == 87) asm volatile("" ::: "s87");
    if (SG == 95) asm volatile("" ::: "s95");
    if (SG == 101) asm volatile("" ::: "s101");
    const u64 t0 = wall_clock64();
    if (threadIdx.x == 0) lds[0] = 1;
    __syncthreads();
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(2); }
    __syncthreads();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t0; st[2 * blockIdx.x + 1] = wall_clock64(); }
}
template <int NT, int SG = 0> static void run(u32 grid, size_t ldsb, const char* name) {
    u64* d; CK(hipMalloc(&d, grid * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin<NT, SG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k_spin<NT, SG>), dim3(grid), dim3(NT), ldsb, 0, d, 2000u); CK(hipDeviceSynchronize()); }
    std::vector<u64> h(grid * 2); CK(hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost));
    u64 t0 = ~0ull, t1 = 0; for (u32 i = 0; i < grid; i++) { t0 = std::min(t0, h[2 * i]); t1 = std::max(t1, h[2 * i + 1]); }
    // workgroups alive 10 us after the first start
    u32 alive = 0; const u64 probe = t0 + 1000; for (u32 i = 0; i < grid; i++) if (h[2 * i] <= probe && h[2 * i + 1] > probe) alive++;
    printf("%-44s grid %5u  total %8.2f us  (spin 20 us)  alive at +10 us: %u\n", name, grid, (t1 - t0) / 100.0, alive);
    CK(hipFree(d));
}
int main() {
    run<1024>(1024, 1024, "1024 thr, 1 KiB LDS");
    run<1024>(1024, 40 << 10, "1024 thr, 40 KiB LDS");
    run<1024>(1024, 60 << 10, "1024 thr, 60 KiB LDS");
    run<1024>(1024, 79 << 10, "1024 thr, 79 KiB LDS");
    run<1024>(1024, 80 << 10, "1024 thr, 80 KiB LDS");
    run<512>(2048, 40 << 10, "512 thr, 40 KiB LDS");
    run<512>(2048, 80 << 10, "512 thr, 80 KiB LDS");
    run<512>(2048, 53 << 10, "512 thr, 53 KiB LDS");
    run<256>(4096, 40 << 10, "256 thr, 40 KiB LDS");
    run<256>(4096, 20 << 10, "256 thr, 20 KiB LDS");
    run<1024, 39>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s39");
    run<1024, 47>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s47");
    run<1024, 55>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s55");
    run<1024, 63>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s63");
    run<1024, 71>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s71");
    run<1024, 79>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s79");
    run<1024, 87>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s87");
    run<1024, 95>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s95");
    run<1024, 101>(1024, 80 << 10, "1024 thr, 80 KiB LDS, SGPRs up to s101");
    run<1024, 95>(1024, 40 << 10, "1024 thr, 40 KiB LDS, SGPRs up to s95");
    return 0;
}
