// atomic_probe.hip — micro-measurements that size the K1 design on MI355X (not part of the product):
// device-scope u64 atomic throughput under different address distributions, LDS atomics, and the
// plain streaming-read rate of 32-byte records.  Build: hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

template <int WORDS, bool RET>
__global__ __launch_bounds__(256) void k_atomic(u64* acc, const u32* idx, u64 n, u64* sink) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        u64* a = acc + (size_t)idx[i] * 4;
#pragma unroll
        for (int w = 0; w < WORDS; w++) { if (RET) s += atomicAdd(&a[w], 1ull + i); else atomicAdd(&a[w], 1ull + i); }
    }
    if (RET && s == 0x1234567) *sink = s;
}
__global__ __launch_bounds__(256) void k_stream(const uint4* p, u64 n, u64* sink) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) { uint4 a = p[2 * i], b = p[2 * i + 1]; s += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w; }
    if (s == 0x1234567) *sink = s;
}
__global__ __launch_bounds__(256) void k_stream16(const uint4* p, u64 n16, u64* sink) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) { uint4 a = p[i]; s += a.x ^ a.y ^ a.z ^ a.w; }
    if (s == 0x1234567) *sink = s;
}
__global__ __launch_bounds__(256) void k_lds(const u32* idx, u64 n, u64* out) {
    __shared__ u64 t[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) t[i] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        u32 k = idx[i] & 1023u;
        atomicAdd(&t[k * 4], 1ull); atomicAdd(&t[k * 4 + 1], i); atomicMax(&t[k * 4 + 2], i); atomicAdd(&t[k * 4 + 3], i * i);
    }
    __syncthreads();
    if (threadIdx.x == 0 && t[5] == 0x1234567) *out = t[5];
}

static double run(const char* name, int reps, std::function<void()> f, double items) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double us = ms * 1000.0 / reps;
    printf("%-44s %10.2f us   %8.3f G items/s\n", name, us, items / us / 1e3);
    return us;
}
#include <functional>
int main() {
    const u64 N = 1 << 20; const u32 M = 50000;
    std::vector<u32> zero(N, 0), uni(N), zipf(N), line(N);
    srand(1);
    std::vector<double> cdf(M); double c = 0; for (u32 r = 0; r < M; r++) { c += pow(r + 1.0, -0.8); cdf[r] = c; }
    std::vector<u32> perm(M); for (u32 i = 0; i < M; i++) perm[i] = i; std::random_shuffle(perm.begin(), perm.end());
    for (u64 i = 0; i < N; i++) {
        uni[i] = (u32)(((u64)rand() * RAND_MAX + rand()) % M);
        double u = (rand() + 0.5) / (RAND_MAX + 1.0) * c;
        zipf[i] = perm[std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()];
        line[i] = i & 1;   // two slots = same 64 B region
    }
    u64 *acc, *sink; u32* idx; uint4* ev;
    CK(hipMalloc(&acc, (size_t)M * 32 + 4096)); CK(hipMemset(acc, 0, (size_t)M * 32 + 4096)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&idx, N * 4));
    const int NB = 12; CK(hipMalloc(&ev, (size_t)NB * N * 32)); CK(hipMemset(ev, 1, (size_t)NB * N * 32));
    auto up = [&](std::vector<u32>& v) { CK(hipMemcpy(idx, v.data(), N * 4, hipMemcpyHostToDevice)); };
    int grid = 2048;
    printf("N = %llu atomics-groups per launch, M = %u slots of 32 B\n", N, M);
    int b = 0;
    run("stream 32B records (lane=record), ring", 24, [&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, ev + (size_t)(b++ % NB) * N * 2, N, sink); }, N);
    run("stream 16B coalesced, ring", 24, [&] { hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(256), 0, 0, ev + (size_t)(b++ % NB) * N * 2, 2 * N, sink); }, N);
    up(zero);
    run("same slot, 1 word, no return", 3, [&] { hipLaunchKernelGGL((k_atomic<1, false>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    run("same slot, 4 words, no return", 3, [&] { hipLaunchKernelGGL((k_atomic<4, false>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    run("same slot, 1 word, returning", 3, [&] { hipLaunchKernelGGL((k_atomic<1, true>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    up(line);
    run("two slots (one 64B line), 4 words", 3, [&] { hipLaunchKernelGGL((k_atomic<4, false>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    up(uni);
    run("uniform over 50k slots, 1 word", 10, [&] { hipLaunchKernelGGL((k_atomic<1, false>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    run("uniform over 50k slots, 4 words", 10, [&] { hipLaunchKernelGGL((k_atomic<4, false>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    run("uniform over 50k slots, 4 words, returning", 10, [&] { hipLaunchKernelGGL((k_atomic<4, true>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    up(zipf);
    run("zipf(0.8) over 50k slots, 1 word", 10, [&] { hipLaunchKernelGGL((k_atomic<1, false>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    run("zipf(0.8) over 50k slots, 4 words", 10, [&] { hipLaunchKernelGGL((k_atomic<4, false>), dim3(grid), dim3(256), 0, 0, acc, idx, N, sink); }, N);
    run("LDS 4 atomics/item zipf&1023", 10, [&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, 0, idx, N, sink); }, N);
    up(zero);
    run("LDS 4 atomics/item same key", 10, [&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, 0, idx, N, sink); }, N);
    // grid sweep for the zipf case
    up(zipf);
    for (int g : {256, 512, 1024, 4096}) { char nm[64]; snprintf(nm, 64, "zipf 4 words, grid=%d", g);
        run(nm, 10, [&] { hipLaunchKernelGGL((k_atomic<4, false>), dim3(g), dim3(256), 0, 0, acc, idx, N, sink); }, N); }
    return 0;
}
