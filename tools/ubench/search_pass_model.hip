// Micro-benchmark: a model of the pipelined search's inner structure, to see which ingredient keeps the
// matrix pipes from being busy when four 4-wave blocks share a CU.  Per "tile" a wave runs three passes
// of 16 LDS row reads (ds_read_b128, two rows ahead) feeding 40 v_mfma_i32_16x16x64_i8 into 8
// accumulators, optionally followed by `glue` dependent-free vector instructions on the accumulators
// (shifts, maxima), optionally with an s_barrier before each pass, optionally with 1 KB LDS-bound
// global loads per wave and pass.  Launched with 1..4 blocks per CU; reports cycles per wave-tile.
//   mode bits: 1 = LDS reads, 2 = glue vector work, 4 = barriers, 8 = global -> LDS loads,
//   16 = the loads re-read the same 12 KB (cache hits), 32 = the loads go to registers instead of LDS
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
#define PITCH 272
#define ROWS 41

__device__ __forceinline__ void gload_lds16(const void* base, uint32_t off, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(off), "s"(base) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_model(const uint8_t* src, int tiles, long long* out, int* sink) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[3][704 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4;
    for (int i = tid; i < 3 * 704 * 4; i += 256) ((int*)ring)[i] = i * 2654435761u;
    __syncthreads();
    const int aoff = (wv * 8 + (g >> 1)) * PITCH + 16 * ((lane & 15) + (g & 1));
    v4i B[5];
    for (int s = 0; s < 5; ++s) B[s] = v4i{lane + s, 2 * s, 3, (int)blockIdx.x};
    v4i acc[8];
    for (int y = 0; y < 8; ++y) acc[y] = v4i{y, y, y, y};
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&ring[0][0] + 1024 * wv);
    uint32_t goff = blockIdx.x * ((uint32_t)tiles * 3u * 12288u) + tid * 16u;   // every block streams its own planes
    int tmax = 0;
    const long long t0 = clock64();
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (MODE & 8) {
                // the slot of the pass before last takes a new plane: 3 loads per thread
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            }
            if (MODE & 4) __builtin_amdgcn_s_barrier();
            if (MODE & 8) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (MODE & 32) {
                        v4i d;
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(goff + 4096u * k), "s"(src) : "memory");
                    } else {
                        gload_lds16(src, goff + 4096u * k, ring_lds + ((j + 2) % 3) * 11264 + 4096 * (k < 2 ? k : 0));
                    }
                }
                if (!(MODE & 16)) goff += 12288u;
            }
            if (MODE & 2) {
                if (j == 0) {
#pragma unroll
                    for (int y = 0; y < 8; ++y) acc[y] = v4i{(tmax >> y) << 20, (tmax >> (y + 1)) << 20, (tmax >> (y + 2)) << 20, (tmax >> (y + 3)) << 20};
                }
                if (j == 1) {
#pragma unroll
                    for (int y = 0; y < 8; ++y) acc[y] <<= 7;
                }
                if (j == 2) {
#pragma unroll
                    for (int y = 0; y < 8; ++y) acc[y] <<= 1;
                }
            }
            const uint8_t* base = &ring[j][0] + aoff;
            v4i Fa, Fb;
            if (MODE & 1) { Fa = *(const v4i*)(base); Fb = *(const v4i*)(base + PITCH); }
            else { Fa = v4i{lane, j, t, 1}; Fb = v4i{lane, j, t, 2}; }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 5; ++s) { const int yy = r - 2 * s; if (yy >= 0 && yy < 8) acc[yy] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fa, B[s], acc[yy], 0, 0, 0); }
                __builtin_amdgcn_sched_barrier(0);
                if ((MODE & 1) && r + 2 < 16) Fa = *(const v4i*)(base + (r + 2) * PITCH);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 5; ++s) { const int yy = r + 1 - 2 * s; if (yy >= 0 && yy < 8) acc[yy] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fb, B[s], acc[yy], 0, 0, 0); }
                __builtin_amdgcn_sched_barrier(0);
                if ((MODE & 1) && r + 3 < 16) Fb = *(const v4i*)(base + (r + 3) * PITCH);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int y = 0; y < 8; ++y)
#pragma unroll
                for (int k = 0; k < 4; ++k) tmax = tmax > acc[y][k] ? tmax : acc[y][k];
        }
    }
    if (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    int s = tmax;
    for (int y = 0; y < 8; ++y) s += acc[y][0] + acc[y][1] + acc[y][2] + acc[y][3];
    if (s == 0x12345678) sink[0] = s;
    if (tid == 0 && blockIdx.x == gridDim.x / 2) out[0] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
int run(const uint8_t* src, long long* d_out, int* d_sink, const char* what) {
    const int tiles = 64;
    printf("%-58s", what);
    for (int per_cu = 1; per_cu <= 4; ++per_cu) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_model<MODE>, dim3(256 * per_cu), dim3(256), 0, 0, src, tiles, d_out, d_sink);
            CK(hipDeviceSynchronize());
        }
        long long h = 0;
        CK(hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost));
        // cycles per wave-tile seen by one SIMD: block time / tiles / blocks per CU
        printf("  %d/CU: %6.0f", per_cu, (double)h / tiles / per_cu);
    }
    printf("   (cycles per wave-tile and SIMD; 1920 = matrix pipe busy)\n");
    return 0;
}

int main() {
    uint8_t* src;
    long long* d_out;
    int* d_sink;
    const size_t bytes = (size_t)1024 * 64 * 3 * 12288;   // 1024 blocks x 64 tiles x 3 planes of 12 KB
    CK(hipMalloc(&src, bytes + (1 << 20)));
    CK(hipMemset(src, 1, bytes + (1 << 20)));
    CK(hipMalloc(&d_out, 64));
    CK(hipMalloc(&d_sink, 64));
    if (run<0>(src, d_out, d_sink, "matrix instructions only")) return 1;
    if (run<1>(src, d_out, d_sink, "+ LDS row reads")) return 1;
    if (run<3>(src, d_out, d_sink, "+ LDS row reads + vector work")) return 1;
    if (run<5>(src, d_out, d_sink, "+ LDS row reads + barriers")) return 1;
    if (run<7>(src, d_out, d_sink, "+ LDS row reads + vector work + barriers")) return 1;
    if (run<13>(src, d_out, d_sink, "+ LDS row reads + barriers + global->LDS loads")) return 1;
    if (run<15>(src, d_out, d_sink, "everything")) return 1;
    if (run<29>(src, d_out, d_sink, "LDS reads + barriers + global->LDS loads, cache hits")) return 1;
    if (run<45>(src, d_out, d_sink, "LDS reads + barriers + global->register loads")) return 1;
    if (run<9>(src, d_out, d_sink, "LDS reads + global->LDS loads, no barriers")) return 1;
    return 0;
}
