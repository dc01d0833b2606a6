// Bitwise check of the DPP / permlane-swap xor butterfly used by k5_edge_score against the
// __shfl_xor reference (strides 32, 16, 8, 4, 2, 1).  Prints "OK" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../alaz_amd/csrc/sg_kernels.h"

__global__ void k(const float* in, float* a, float* b, unsigned* xa, unsigned* xb) {
    const unsigned t = threadIdx.x;
    float r = in[blockIdx.x * 64 + t], s = r;
#pragma unroll
    for (int st = 32; st >= 1; st >>= 1) r = r + __shfl_xor(r, st, 64);
    s = wave_butterfly_sum_f32(s);
    a[blockIdx.x * 64 + t] = r; b[blockIdx.x * 64 + t] = s;
    // partner maps, stride by stride, on lane ids
    unsigned v = t;
    xa[blockIdx.x * 64 + t] = 0; xb[blockIdx.x * 64 + t] = 0;
    if (blockIdx.x < 6) {
        const int st = 32 >> blockIdx.x;
        xa[blockIdx.x * 64 + t] = (unsigned)__shfl_xor((int)v, st, 64);
        xb[blockIdx.x * 64 + t] = __float_as_uint(xor_partner_f32(__uint_as_float(v), st));
    }
}

int main() {
    const int NB = 4096, N = NB * 64;
    float* h = (float*)malloc(N * 4);
    srand(7);
    for (int i = 0; i < N; i++) { unsigned u = ((unsigned)rand() << 9) ^ (unsigned)rand(); u = (u & 0x807FFFFFu) | ((100u + (rand() % 56)) << 23); memcpy(&h[i], &u, 4); }
    float *din, *da, *db; unsigned *xa, *xb;
    hipMalloc(&din, N * 4); hipMalloc(&da, N * 4); hipMalloc(&db, N * 4); hipMalloc(&xa, N * 4); hipMalloc(&xb, N * 4);
    hipMemcpy(din, h, N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, din, da, db, xa, xb);
    float* ra = (float*)malloc(N * 4); float* rb = (float*)malloc(N * 4); unsigned* pa = (unsigned*)malloc(N * 4); unsigned* pb = (unsigned*)malloc(N * 4);
    hipMemcpy(ra, da, N * 4, hipMemcpyDeviceToHost); hipMemcpy(rb, db, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(pa, xa, N * 4, hipMemcpyDeviceToHost); hipMemcpy(pb, xb, N * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 6 * 64; i++) if (pa[i] != pb[i]) { printf("PARTNER MISMATCH stride %d lane %d: %u vs %u\n", 32 >> (i / 64), i % 64, pa[i], pb[i]); return 1; }
    for (int i = 0; i < N; i++) if (memcmp(&ra[i], &rb[i], 4)) { printf("SUM MISMATCH at %d: %a vs %a\n", i, ra[i], rb[i]); return 1; }
    printf("OK %d waves\n", NB);
    return 0;
}
