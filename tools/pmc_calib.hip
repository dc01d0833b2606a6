// pmc_calib.hip — known-byte-count kernels in the access patterns of the two K1 kernels, to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on this chip (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern before trusting an
// absolute").  Not part of the product.  Every kernel moves BYTES bytes of a buffer larger than the 256 MiB Infinity Cache, once:
//   cal_read16_stream   pass A's event stream: 16 bytes per lane, two per 32-byte event, coalesced
//   cal_read_pieces     pass B's record reads: 4 lanes x 16 bytes side by side per piece, pieces 3 KiB apart (the first 64 bytes of each)
//   cal_write_runs64    pass A's copy-out: 64-byte runs (8 lanes x 8 bytes) at piece-strided addresses
//   cal_write_rows44    pass B's compaction: per edge 4 + 4 + 32 + 4 bytes into four arrays at consecutive slots
//   cal_write16_stream  a plain coalesced 16-byte-per-lane write (the reference for WRITE_SIZE)
// Build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE
// (tools/gpu.sh calib:TAG folds both into profiles/TAG_pmc_calibration.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(1024) void cal_read16_stream(const uint4* p, u64 n16, u64* sink) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (u64)gridDim.x * 1024) { const uint4 a = p[i]; s += a.x ^ a.y ^ a.z ^ a.w; }
    if (s == 0x1234567) *sink = s;
}
// piece q (3072 bytes): lanes 4 q .. 4 q + 3 read its first 64 bytes
__global__ __launch_bounds__(1024) void cal_read_pieces(const uint4* p, u64 npieces, u64* sink) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < npieces * 4; i += (u64)gridDim.x * 1024) { const uint4 a = p[(i >> 2) * 192 + (i & 3)]; s += a.x ^ a.y ^ a.z ^ a.w; }
    if (s == 0x1234567) *sink = s;
}
// run r (64 bytes = 8 records) goes to piece r at a varying 64-byte-aligned... no: 8-byte-aligned position, as pass A's runs do
__global__ __launch_bounds__(1024) void cal_write_runs64(u64* p, u64 nruns) {
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < nruns * 8; i += (u64)gridDim.x * 1024) {
        const u64 r = i >> 3, j = i & 7;
        p[r * 384 + ((r * 5) % 300) + j] = i;                        // piece of 384 units, the run at position (5 r mod 300): not line-aligned
    }
}
__global__ __launch_bounds__(1024) void cal_write_rows44(u32* ef, u32* et, ulonglong2* acc, u32* er, u64 nrows) {
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < nrows; i += (u64)gridDim.x * 1024) {
        // thread t of workgroup b writes slot b * 2048 + (t mixed): a workgroup's rows land in its own 2048-slot partition in arrival order
        const u64 slot = (i / 1024) * 1024 + ((i * 37) % 1024);
        ef[slot] = (u32)i; et[slot] = (u32)(i >> 3); acc[2 * slot] = make_ulonglong2(i, i + 1); acc[2 * slot + 1] = make_ulonglong2(i + 2, i + 3); er[slot] = (u32)(i & 7);
    }
}
__global__ __launch_bounds__(1024) void cal_write16_stream(uint4* p, u64 n16) {
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (u64)gridDim.x * 1024) p[i] = make_uint4((u32)i, 1u, 2u, 3u);
}

int main() {
    const u64 BYTES = 384ull << 20;                                  // moved per launch
    void* buf; u64* sink;
    CK(hipMalloc(&buf, 3ull << 30)); CK(hipMemset(buf, 1, 3ull << 30)); CK(hipMalloc(&sink, 64));
    char* b = (char*)buf;
    const int G = 512, REP = 4;
    printf("bytes per launch: read16_stream %llu, read_pieces %llu, write_runs64 %llu, write_rows44 %llu, write16_stream %llu\n",
           BYTES, (BYTES / 3072) * 64, (BYTES / 3072) * 64, (BYTES / 44) * 44, BYTES);
    for (int r = 0; r < REP; r++) {                                  // (every launch of a kernel touches a different 768 MiB region: nothing is cache-resident)
        char* reg = b + (u64)(r % 3) * (1ull << 30);
        hipLaunchKernelGGL(cal_read16_stream, dim3(G), dim3(1024), 0, 0, (const uint4*)reg, BYTES / 16, sink);
        hipLaunchKernelGGL(cal_read_pieces, dim3(G), dim3(1024), 0, 0, (const uint4*)reg, BYTES / 3072, sink);          // the pieces of a 384 MiB slab, 64 bytes each
        hipLaunchKernelGGL(cal_write_runs64, dim3(G), dim3(1024), 0, 0, (u64*)reg, BYTES / 3072);
        hipLaunchKernelGGL(cal_write_rows44, dim3(G), dim3(1024), 0, 0, (u32*)reg, (u32*)(reg + (64ull << 20)), (ulonglong2*)(reg + (128ull << 20)), (u32*)(reg + (640ull << 20)), BYTES / 44);
        hipLaunchKernelGGL(cal_write16_stream, dim3(G), dim3(1024), 0, 0, (uint4*)reg, BYTES / 16);
        CK(hipDeviceSynchronize());
    }
    printf("done\n");
    return 0;
}
