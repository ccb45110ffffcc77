// Micro-benchmark: cost of one dependent kernel boundary inside a replayed hipGraph on this GPU.
// A chain of N kernels per graph, each of `blocks` blocks that do almost nothing (one dependent
// global load + store), replayed R times; prints microseconds per kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

__global__ void k_tiny(int* p, int hops) {
    // `hops` dependent global round trips by thread 0 of each block
    if (threadIdx.x == 0) {
        int v = blockIdx.x & 63;
        for (int i = 0; i < hops; ++i) v = __hip_atomic_load(p + (v & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 12345) p[64] = v;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    int* d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int configs[][3] = {{1, 256, 0}, {512, 256, 0}, {2048, 64, 0}, {1, 256, 1}, {1, 256, 2}, {1, 256, 4}, {1, 256, 8}};
    for (auto& c : configs) {
        const int N = 256, R = 20;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(c[0]), dim3(c[1]), 0, s, d, c[2]);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("graph: %4d blocks x %3d threads, %d dependent loads: %.2f us per kernel\n", c[0], c[1], c[2], us / (N * R));
        // the same chain launched eagerly
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r)
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(c[0]), dim3(c[1]), 0, s, d, c[2]);
        CK(hipStreamSynchronize(s));
        const double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("eager: %4d blocks x %3d threads, %d dependent loads: %.2f us per kernel\n", c[0], c[1], c[2], us2 / (N * R));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
