// Micro-benchmark: issue rate of v_dot4 (8-bit dot product, int32 accumulate) on gfx950,
// with the second operand in an SGPR as the search kernel uses it.  Prints lane-ops/s and MAC/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NACC, bool UNSIGNED>
__global__ __launch_bounds__(256) void k_dot4(int* out, const int* tm, int iters) {
    int acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = threadIdx.x + k;
    int a = threadIdx.x * 0x01010101 + blockIdx.x;
    typedef const int __attribute__((address_space(4)))* cptr;
    cptr t = (cptr)(unsigned long long)tm;
    const int t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (UNSIGNED) {
                acc[k] = (int)__builtin_amdgcn_udot4((unsigned)a, (unsigned)t0, (unsigned)acc[k], false);
                acc[k] = (int)__builtin_amdgcn_udot4((unsigned)a, (unsigned)t1, (unsigned)acc[k], false);
                acc[k] = (int)__builtin_amdgcn_udot4((unsigned)a, (unsigned)t2, (unsigned)acc[k], false);
                acc[k] = (int)__builtin_amdgcn_udot4((unsigned)a, (unsigned)t3, (unsigned)acc[k], false);
            } else {
                acc[k] = __builtin_amdgcn_sdot4(a, t0, acc[k], false);
                acc[k] = __builtin_amdgcn_sdot4(a, t1, acc[k], false);
                acc[k] = __builtin_amdgcn_sdot4(a, t2, acc[k], false);
                acc[k] = __builtin_amdgcn_sdot4(a, t3, acc[k], false);
            }
        }
        a += 0x01020304;
    }
    int s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mad(int* out, const int* tm, int iters) {
    int acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = threadIdx.x + k;
    int a = threadIdx.x * 0x01010101 + blockIdx.x;
    typedef const int __attribute__((address_space(4)))* cptr;
    cptr t = (cptr)(unsigned long long)tm;
    const int t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            acc[k] = a * t0 + acc[k];
            acc[k] = a * t1 + acc[k];
            acc[k] = a * t2 + acc[k];
            acc[k] = a * t3 + acc[k];
        }
        a += 0x01020304;
    }
    int s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8, iters = 2000;
    int *out, *tm; hipMalloc(&out, blocks * 256 * sizeof(int)); hipMalloc(&tm, 64);
    int h[4] = {0x01020304, 0x7f010203, 0x00010001, 0x01000100}; hipMemcpy(tm, h, 16, hipMemcpyHostToDevice);
    constexpr int NACC = 16;
    const double ops = (double)blocks * 256 * iters * NACC * 4;
    double ms = time_ms([&] { hipLaunchKernelGGL((k_dot4<NACC, false>), dim3(blocks), dim3(256), 0, 0, out, tm, iters); });
    printf("sdot4 (i8, SGPR operand): %.3f ms  %.2f T lane-ops/s  %.2f T MAC/s\n", ms, ops / ms / 1e9, 4 * ops / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_dot4<NACC, true>), dim3(blocks), dim3(256), 0, 0, out, tm, iters); });
    printf("udot4 (u8, SGPR operand): %.3f ms  %.2f T lane-ops/s  %.2f T MAC/s\n", ms, ops / ms / 1e9, 4 * ops / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_mad<NACC>), dim3(blocks), dim3(256), 0, 0, out, tm, iters); });
    printf("int32 mad (v_mad_u32_u24/v_mul_lo+add): %.3f ms  %.2f T lane-ops/s\n", ms, ops / ms / 1e9);
    return 0;
}
