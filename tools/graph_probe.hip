// graph_probe.hip — what a chain of DEPENDENT kernel launches costs on this box, as stream launches and as one captured hipGraph.
//   hipcc --offload-arch=gfx950 -O2 -o tools/graph_probe tools/graph_probe.hip && tools/graph_probe
// Sixteen kernels per "window", each reading a flag the one before wrote (the shape of the window close: every launch depends on the last);
// three bodies: empty (reads one word and returns), 1 KiB of by-value arguments like the engine's Dev, and 20 us of spinning.
// Reports us per kernel from hipEvents around 200 windows.  (is the close worth capturing as a graph?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { unsigned long long w[128]; };
__global__ void k_flag(unsigned* f) { if (threadIdx.x == 0 && blockIdx.x == 0) f[0] = f[0] + 1; }
__global__ void k_big(Big b, unsigned* f) { if (threadIdx.x == 0 && blockIdx.x == 0) f[0] = f[0] + (unsigned)b.w[5]; }
__global__ void k_spin(unsigned* f, unsigned ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) f[0] = f[0] + 1;
}
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 1; } } while (0)
int main() {
    unsigned* f; CK(hipMalloc(&f, 64)); CK(hipMemset(f, 0, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    Big big = {};
    const int NK = 16, NW = 200;
    for (int body = 0; body < 3; body++) for (int grid : {1, 1024}) {
        auto enqueue = [&]() {
            for (int k = 0; k < NK; k++) {
                if (body == 0) hipLaunchKernelGGL(k_flag, dim3(grid), dim3(256), 0, s, f);
                else if (body == 1) hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, s, big, f);
                else hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, s, f, 2000u);     // 100 MHz ticks: 20 us
            }
        };
        for (int w = 0; w < 20; w++) enqueue();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); for (int w = 0; w < NW; w++) enqueue(); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms_stream; CK(hipEventElapsedTime(&ms_stream, a, b));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); enqueue(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 20; w++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); for (int w = 0; w < NW; w++) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms_graph; CK(hipEventElapsedTime(&ms_graph, a, b));
        printf("%-28s grid %5d: stream %6.2f us per kernel, graph %6.2f us per kernel\n",
               body == 0 ? "flag kernel" : body == 1 ? "1 KiB by-value argument" : "20 us of work", grid, ms_stream * 1e3 / (NK * NW), ms_graph * 1e3 / (NK * NW));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
