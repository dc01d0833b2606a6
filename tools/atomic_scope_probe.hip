// atomic_scope_probe.hip — does a narrower-than-agent atomic scope on per-XCD replicas run in the XCD's own L2,
// and how fast?  (Design probe for K1; not part of the product.)  Each workgroup adds into replica[XCC_ID],
// so all writers of one replica share one L2.  Checks the totals afterwards: lost updates => unusable.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <functional>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ u32 xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7u; }

template <int SCOPE, int WORDS, bool REPL>
__global__ __launch_bounds__(256) void k_atomic(u64* acc, const u32* idx, u64 n, u32 M) {
    u64* base = REPL ? acc + (size_t)xcc_id() * M * 4 : acc;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        u64* a = base + (size_t)idx[i] * 4;
#pragma unroll
        for (int w = 0; w < WORDS; w++) __hip_atomic_fetch_add(&a[w], 1ull, __ATOMIC_RELAXED, SCOPE);
    }
}
__global__ void k_xcc(u32* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

static double run(const char* name, int reps, std::function<void()> f, double items) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double us = ms * 1000.0 / reps;
    printf("%-58s %10.2f us   %8.3f G items/s\n", name, us, items / us / 1e3);
    return us;
}
int main() {
    const u64 N = 1 << 20; const u32 M = 50000;
    std::vector<u32> uni(N), zipf(N);
    srand(1);
    std::vector<double> cdf(M); double c = 0; for (u32 r = 0; r < M; r++) { c += pow(r + 1.0, -0.8); cdf[r] = c; }
    std::vector<u32> perm(M); for (u32 i = 0; i < M; i++) perm[i] = i; std::random_shuffle(perm.begin(), perm.end());
    for (u64 i = 0; i < N; i++) {
        uni[i] = (u32)(((u64)rand() * RAND_MAX + rand()) % M);
        double u = (rand() + 0.5) / (RAND_MAX + 1.0) * c;
        zipf[i] = perm[std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()];
    }
    u64* acc; u32* idx; u32* xo;
    const size_t accb = (size_t)8 * M * 32;
    CK(hipMalloc(&acc, accb)); CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&xo, 64 * 4));
    hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, 0, xo); std::vector<u32> hx(64); CK(hipMemcpy(hx.data(), xo, 64 * 4, hipMemcpyDeviceToHost));
    printf("XCC id of blocks 0..15:"); for (int i = 0; i < 16; i++) printf(" %u", hx[i]); printf("\n");
    auto check = [&](const char* what, int reps_total, int words) {
        std::vector<u64> h(accb / 8); CK(hipMemcpy(h.data(), acc, accb, hipMemcpyDeviceToHost));
        u64 s = 0; for (u64 v : h) s += v;
        printf("    check %-40s sum=%llu expected=%llu %s\n", what, s, (u64)N * words * reps_total, s == (u64)N * words * reps_total ? "OK" : "LOST UPDATES");
    };
    int grid = 2048;
    for (int dist = 0; dist < 2; dist++) {
        CK(hipMemcpy(idx, dist ? zipf.data() : uni.data(), N * 4, hipMemcpyHostToDevice));
        const char* dn = dist ? "zipf(0.8)" : "uniform";
        char nm[96];
#define CASE(SC, SCN, REPL) { CK(hipMemset(acc, 0, accb)); snprintf(nm, 96, "%s 4 words scope=%s %s", dn, SCN, REPL ? "per-XCD replica" : "shared"); \
        run(nm, 4, [&] { hipLaunchKernelGGL((k_atomic<SC, 4, REPL>), dim3(grid), dim3(256), 0, 0, acc, idx, N, M); }, N); CK(hipDeviceSynchronize()); check(nm, 5, 4); }
        CASE(__HIP_MEMORY_SCOPE_AGENT, "agent", false)
        CASE(__HIP_MEMORY_SCOPE_AGENT, "agent", true)
        CASE(__HIP_MEMORY_SCOPE_WORKGROUP, "workgroup", true)
        CASE(__HIP_MEMORY_SCOPE_WAVEFRONT, "wavefront", true)
        CASE(__HIP_MEMORY_SCOPE_WORKGROUP, "workgroup", false)
    }
    return 0;
}
