// Micro-benchmark: do matrix-core instructions of one wave overlap with vector-ALU instructions of
// another wave on the same SIMD?  One block of 256 or 512 threads per CU; waves 0..3 (one per SIMD)
// run a loop of independent v_mfma_i32_16x16x64_i8, waves 4..7 (the second wave of each SIMD) run a
// loop of independent 32-bit integer multiply-adds.  Each mode is timed with the wall clock inside
// the kernel by wave 0 / wave 4: MFMA waves alone, VALU waves alone, both together.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef int v4i __attribute__((ext_vector_type(4)));

// mode bit 0: waves 0..3 run the MFMA loop; bit 1: waves 4..7 run the VALU loop
__global__ __launch_bounds__(512) void k_overlap(int mode, int iters, long long* out, int* sink) {
    const int wv = threadIdx.x >> 6;
    const bool mf = wv < 4 && (mode & 1);
    const bool va = wv >= 4 && (mode & 2);
    __syncthreads();
    const long long t0 = wall_clock64();
    if (mf) {
        v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
        v4i acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = v4i{k, k, k, k};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
        }
        int s = 0;
        for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
        if (s == 0x12345678) sink[0] = s;
    }
    if (va) {
        int x[16];
        for (int k = 0; k < 16; ++k) x[k] = threadIdx.x + k;
        const int m = blockIdx.x * 2 + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = (x[k] ^ (x[k] >> 3)) + m;   // full-rate integer operations only
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = (x[k] ^ (x[k] << 5)) - k;
        }
        int s = 0;
        for (int k = 0; k < 16; ++k) s += x[k];
        if (s == 0x12345678) sink[1] = s;
    }
    const long long t1 = wall_clock64();
    if (blockIdx.x == 7 && (threadIdx.x == 0 || threadIdx.x == 256)) out[wv >> 2] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    long long* d_out;
    int* d_sink;
    CK(hipMalloc(&d_out, 64));
    CK(hipMalloc(&d_sink, 64));
    const int iters = 20000;
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(d_out, 0, 64));
            hipLaunchKernelGGL(k_overlap, dim3(256), dim3(512), 0, 0, mode, iters, d_out, d_sink);
            CK(hipDeviceSynchronize());
        }
        long long h[2];
        CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
        // wall clock: 100 MHz
        printf("mode %d (%s): MFMA wave %.1f us (%.2f ns per MFMA), VALU wave %.1f us (%.2f ns per 1/96 of an iteration)\n", mode,
               mode == 1 ? "MFMA alone" : mode == 2 ? "VALU alone" : "both on each SIMD",
               h[0] * 0.01, h[0] * 10.0 / (iters * 8.0), h[1] * 0.01, h[1] * 10.0 / (iters * 96.0));
    }
    return 0;
}
