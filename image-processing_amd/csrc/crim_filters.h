// HIP kernels of the noise-filter stage that follows the inpainter in the reference's pipeline:
// filter_noise, /root/reference/src/utils.py:106-138 -- cv2.medianBlur (:118),
// cv2.fastNlMeansDenoisingColored(h, hColor=1, template 9, search 31) (:124-126), Laplacian
// sharpening with cv2.filter2D(CV_8U) and cv2.subtract (:128-136).  All arithmetic is integer
// (OpenCV's 8-bit paths are fixed point), so results are compared bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NF_TW 9                 // template window of the non-local means (utils.py:126)
#define NF_SW 31                // search window
#define NF_TWH (NF_TW / 2)
#define NF_SWH (NF_SW / 2)
#define NF_BORDER (NF_TWH + NF_SWH)
#define NF_TILE 16              // output tile edge of the non-local-means kernel
#define NF_REGION (NF_TILE + 2 * NF_BORDER)   // 54: staged source region edge
#define NF_PITCH 80             // LDS pitch in dwords: 16 mod 32, so the 4 rows a wave reads hit each bank twice
#define NF_WLDS 1024            // weights kept in LDS (the table's non-zero prefix is shorter for h <= 15)

struct NfTables {
    const uint16_t* cbrt_tab;   // 3072 entries, shared with the inpainter's BGR -> Lab
    const uint16_t* yf_tab;     // 256 x (y, f(y)) in units of 1/16384
    int coef_fwd[9];            // BGR -> XYZ / white point, 12 bit
    int coef_inv[9];            // XYZ -> BGR * white point, 12 bit
};

// BORDER_REFLECT_101 for any distance (OpenCV's borderInterpolate loops the same way).
__device__ __forceinline__ int nf_reflect101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - i : i;
}
__device__ __forceinline__ int nf_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Median of k x k (k odd, <= 7) per channel with BORDER_REPLICATE, then linear BGR -> Lab
// (COLOR_LBGR2Lab: the integer pipeline of the inpainter's Lab with the gamma table i * 8).
// Writes the L plane and the interleaved (a, b) plane.
template <int K>
__global__ __launch_bounds__(256) void k_nf_median_lab(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ lpl,
                                                       uint8_t* __restrict__ abpl, int H, int W, NfTables t) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const size_t HW = (size_t)H * W;
    const uint8_t* img = bgr + (size_t)blockIdx.z * HW * 3;
    int med[3];
    if (K == 1) {
        const uint8_t* px = img + ((size_t)y * W + x) * 3;
        med[0] = px[0]; med[1] = px[1]; med[2] = px[2];
    } else {
        constexpr int R = K / 2, N = K * K;
        uint8_t v[3][N];
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int yy = nf_clampi(y + i - R, 0, H - 1), xx = nf_clampi(x + j - R, 0, W - 1);
                const uint8_t* px = img + ((size_t)yy * W + xx) * 3;
                v[0][i * K + j] = px[0]; v[1][i * K + j] = px[1]; v[2][i * K + j] = px[2];
            }
        // the median is the smallest value with at least (N+1)/2 samples at or below it
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int lo = 0, hi = 255;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                int cnt = 0;
#pragma unroll
                for (int e = 0; e < N; ++e) cnt += v[c][e] <= mid ? 1 : 0;
                if (cnt >= (N + 1) / 2) hi = mid; else lo = mid + 1;
            }
            med[c] = lo;
        }
    }
    const int b = med[0] * 8, g = med[1] * 8, r = med[2] * 8;
    const int fX = t.cbrt_tab[(b * t.coef_fwd[0] + g * t.coef_fwd[1] + r * t.coef_fwd[2] + 2048) >> 12];
    const int fY = t.cbrt_tab[(b * t.coef_fwd[3] + g * t.coef_fwd[4] + r * t.coef_fwd[5] + 2048) >> 12];
    const int fZ = t.cbrt_tab[(b * t.coef_fwd[6] + g * t.coef_fwd[7] + r * t.coef_fwd[8] + 2048) >> 12];
    const int L = (296 * fY - 1336934 + 16384) >> 15;
    const int a = (500 * (fX - fY) + 128 * 32768 + 16384) >> 15;
    const int bb = (200 * (fY - fZ) + 128 * 32768 + 16384) >> 15;
    const size_t q = (size_t)blockIdx.z * HW + (size_t)y * W + x;
    lpl[q] = (uint8_t)nf_clampi(L, 0, 255);
    abpl[2 * q] = (uint8_t)nf_clampi(a, 0, 255);
    abpl[2 * q + 1] = (uint8_t)nf_clampi(bb, 0, 255);
}

typedef short nf_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ nf_v2s nf_as_v2s(uint32_t v) { return __builtin_bit_cast(nf_v2s, v); }

// 8-bit non-local means (OpenCV's FastNlMeansDenoisingInvoker, restated): for every pixel the
// weighted mean of the 31 x 31 pixels around it, the weight of a candidate being a table look-up of
// (sum of squared differences of the 9 x 9 windows around pixel and candidate) >> shift.
//
// A block produces a 16 x 16 tile from the 54 x 54 source region it stages in LDS (image extended
// with BORDER_REFLECT_101).  Every LDS element is a dword of two 16-bit lanes so that one packed
// subtraction and one v_dot2_i32_i16 handle two squared differences:
//   CH == 2 (the (a, b) image): lane 0 = a, lane 1 = b of one pixel; a window is 9 x 9 dwords;
//   CH == 1 (L): lane 0 = L(y, x), lane 1 = L(y+1, x) (array V) -- a window is 9 columns of
//     4 such pairs plus a last row taken from array Z (lane 1 = 0).
// The pixel's own window stays in registers, and so does the candidate window, which slides along a
// row of the search area one column at a time.
template <int CH>
__global__ __launch_bounds__(256) void k_nf_nlm(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                                const int* __restrict__ wtab, int nw, int shift) {
    constexpr int ROWS = CH == 2 ? NF_TW : (NF_TW + 1) / 2;
    __shared__ uint32_t sV[NF_REGION * NF_PITCH];
    __shared__ uint32_t sZ[CH == 1 ? NF_REGION * NF_PITCH : 1];
    __shared__ int s_w[NF_WLDS];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const size_t HW = (size_t)H * W;
    const uint8_t* img = src + (size_t)blockIdx.z * HW * CH;
    const int x0 = blockIdx.x * NF_TILE - NF_BORDER, y0 = blockIdx.y * NF_TILE - NF_BORDER;
    for (int e = tid; e < NF_WLDS; e += 256) s_w[e] = e < nw ? wtab[e] : 0;
    for (int e = tid; e < NF_REGION * NF_REGION; e += 256) {
        const int ry = e / NF_REGION, rx = e - ry * NF_REGION;
        const size_t q = (size_t)nf_reflect101(y0 + ry, H) * W + nf_reflect101(x0 + rx, W);
        if (CH == 2) sV[ry * NF_PITCH + rx] = (uint32_t)img[2 * q] | ((uint32_t)img[2 * q + 1] << 16);
        else sZ[ry * NF_PITCH + rx] = img[q];
    }
    __syncthreads();
    if (CH == 1) {
        for (int e = tid; e < NF_REGION * NF_REGION; e += 256) {
            const int ry = e / NF_REGION, rx = e - ry * NF_REGION;
            const uint32_t lo = sZ[ry * NF_PITCH + rx];
            const uint32_t hi = ry + 1 < NF_REGION ? sZ[(ry + 1) * NF_PITCH + rx] : 0u;
            sV[ry * NF_PITCH + rx] = lo | (hi << 16);
        }
        __syncthreads();
    }
    // element (r, j) of the window whose top-left corner is LDS position (py, px)
    auto elem = [&](int py, int px, int r, int j) -> uint32_t {
        if (CH == 2) return sV[(py + r) * NF_PITCH + px + j];
        if (r < ROWS - 1) return sV[(py + 2 * r) * NF_PITCH + px + j];
        return sZ[(py + 2 * r) * NF_PITCH + px + j];
    };
    nf_v2s T[ROWS][NF_TW];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int j = 0; j < NF_TW; ++j) T[r][j] = nf_as_v2s(elem(ty + NF_SWH, tx + NF_SWH, r, j));
    uint32_t est0 = 0, est1 = 0, wsum = 0;
    // Candidate windows of one row of the search area (dy fixed, dx = 0 .. 30) overlap in 8 of their 9
    // columns, so the window lives in registers as 9 column slots and every step of dx replaces one
    // column (ROWS LDS reads instead of 9 * ROWS).  Slot k holds window column k at dx = 0 and is the
    // one replaced after step dx when dx % 9 == k: at step dx, window column j sits in slot
    // (j + dx) % 9 -- a compile-time index once the dx loop is unrolled by 9.
    nf_v2s Wn[ROWS][NF_TW];
    auto load_column = [&](int dy, int col, nf_v2s* dst_col /* stride NF_TW */) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) dst_col[r * NF_TW] = nf_as_v2s(elem(ty + dy, tx + col, r, 0));
    };
    auto account = [&](int dy, int dx, int dist) {
        const int a = dist >> shift;
        const uint32_t w = (uint32_t)(a < NF_WLDS ? s_w[a] : (a < nw ? wtab[a] : 0));
        const int cy = ty + dy + NF_TWH, cx = tx + dx + NF_TWH;
        if (CH == 2) {
            const uint32_t v = sV[cy * NF_PITCH + cx];
            est0 += w * (v & 0xffffu);
            est1 += w * (v >> 16);
        } else {
            est0 += w * sZ[cy * NF_PITCH + cx];
        }
        wsum += w;
    };
#pragma unroll 1
    for (int dy = 0; dy < NF_SW; ++dy) {
#pragma unroll
        for (int j = 0; j < NF_TW; ++j) load_column(dy, j, &Wn[0][j]);
#pragma unroll 1
        for (int dx0 = 0; dx0 < NF_SW; dx0 += NF_TW) {
#pragma unroll
            for (int k = 0; k < NF_TW; ++k) {
                const int dx = dx0 + k;
                if (dx < NF_SW) {            // the last group is shorter (31 = 3 * 9 + 4)
                    int dist = 0;
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
#pragma unroll
                        for (int j = 0; j < NF_TW; ++j) {
                            const nf_v2s d = T[r][j] - Wn[r][(j + k) % NF_TW];
                            dist = __builtin_amdgcn_sdot2(d, d, dist, false);
                        }
                    account(dy, dx, dist);
                    // window column 0 of this step is not part of the next one: its slot takes the
                    // column that enters (window column 8 of step dx + 1 = region column dx + 9)
                    if (dx + 1 < NF_SW) load_column(dy, dx + NF_TW, &Wn[0][k]);
                }
            }
        }
    }
    const int ox = blockIdx.x * NF_TILE + tx, oy = blockIdx.y * NF_TILE + ty;
    if (ox < W && oy < H) {
        uint8_t* out = dst + ((size_t)blockIdx.z * HW + (size_t)oy * W + ox) * CH;
        const uint32_t v0 = (est0 + wsum / 2) / wsum;
        out[0] = (uint8_t)(v0 > 255u ? 255u : v0);
        if (CH == 2) {
            const uint32_t v1 = (est1 + wsum / 2) / wsum;
            out[1] = (uint8_t)(v1 > 255u ? 255u : v1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same non-local means in correlation form (the default): with T the pixel's window and W a
// candidate's,  sum (T - W)^2 = sum T^2 + sum W^2 - 2 sum T W  -- exact in integers -- where
//   * sum W^2 over a 9 x 9 window is one look-up in a box-filtered image of squares, built once per tile
//     (and sum T^2 is the same image at the pixel itself);
//   * sum T W is a dot product of BYTES: v_dot4_u32_u8 takes four pixels per instruction, 27 per window and
//     channel instead of 81 subtract / multiply-add pairs.
// Byte operands must line up: a window starts at any column, an LDS dword at a multiple of four.  The
// pixel's own window is therefore kept in registers four times, shifted by 0..3 bytes inside three
// dwords per row (zero elsewhere), and the lanes of a wave take the columns x = 4 lx + w (w = wave
// index): the shift (w + dx) & 3 of a search step is then the same for the whole wave -- a compile-time
// register choice once the walk along a search row is unrolled by four -- and the lanes' candidate
// dwords are consecutive in LDS.  The three candidate dwords per row live in registers and move on by
// one dword every fourth step.
#define NF2_TW 128                         // tile: 128 columns (4 waves x 32 lanes, interleaved) ...
#define NF2_TH 8                           // ... x 8 rows (2 per pass of a wave)
#define NF2_RW (NF2_TW + 2 * NF_BORDER)    // 166: staged region
#define NF2_RH (NF2_TH + 2 * NF_BORDER)    // 46
#define NF2_PITCH 176                      // bytes per staged row (the last candidate dword ends at byte 167)
#define NF2_BW (NF2_RW - NF_TW + 1)        // 158 window positions per row ...
#define NF2_BH (NF2_RH - NF_TW + 1)        // ... and 38 rows of them
#define NF2_BP 40                          // box-sum pitch: positions of one column class (x & 3) per row

template <int CH>
__global__ __launch_bounds__(256) void k_nf_nlm_corr(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                                     const int* __restrict__ wtab, int nw, int shift) {
    __shared__ __attribute__((aligned(16))) uint8_t sP[CH][NF2_RH][NF2_PITCH];   // the staged planes
    // sum over channels of squares, 9 x 9 box sums: [column class x & 3][row][x >> 2]; rows up to NF2_RH
    // during construction (horizontal sums first, vertical ones in place)
    __shared__ uint32_t sB[4][NF2_RH][NF2_BP];
    __shared__ int s_w[NF_WLDS];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lx = lane & 31, ly = lane >> 5;
    const size_t HW = (size_t)H * W;
    const uint8_t* img = src + (size_t)blockIdx.z * HW * CH;
    const int x0 = blockIdx.x * NF2_TW - NF_BORDER, y0 = blockIdx.y * NF2_TH - NF_BORDER;
    for (int e = tid; e < NF_WLDS; e += 256) s_w[e] = e < nw ? wtab[e] : 0;
    for (int e = tid; e < NF2_RH * NF2_PITCH; e += 256) {
        const int ry = e / NF2_PITCH, rx = e - ry * NF2_PITCH;
        const size_t q = (size_t)nf_reflect101(y0 + ry, H) * W + nf_reflect101(x0 + imin(rx, NF2_RW - 1), W);
#pragma unroll
        for (int c = 0; c < CH; ++c) sP[c][ry][rx] = rx < NF2_RW ? img[CH * q + c] : (uint8_t)0;
    }
    __syncthreads();
    // horizontal 9-sums of the squares
    for (int e = tid; e < NF2_RH * NF2_BW; e += 256) {
        const int ry = e / NF2_BW, rx = e - ry * NF2_BW;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < NF_TW; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const uint32_t v = sP[c][ry][rx + j];
                acc += v * v;
            }
        sB[rx & 3][ry][rx >> 2] = acc;
    }
    __syncthreads();
    // vertical 9-sums, in place: a thread owns a column and walks down it (row y only loses a value no
    // later row needs)
    if (tid < NF2_BW) {
        uint32_t* col = &sB[tid & 3][0][tid >> 2];
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < NF_TW; ++i) run += col[i * NF2_BP];
        for (int ry = 0; ry < NF2_BH; ++ry) {
            const uint32_t top = col[ry * NF2_BP];
            col[ry * NF2_BP] = run;
            if (ry + 1 < NF2_BH) run += col[(ry + NF_TW) * NF2_BP] - top;
        }
    }
    __syncthreads();

    const int tx = 4 * lx + w;                    // this lane's column of the tile
    for (int pass = 0; pass < NF2_TH / 2; ++pass) {
        const int ty = 2 * pass + ly;             // ... and row
        // the pixel's own window, shifted by s = 0..3 bytes inside three dwords per row
        uint32_t T[CH][4][NF_TW][3];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int r = 0; r < NF_TW; ++r) {
                uint32_t b[NF_TW];
#pragma unroll
                for (int j = 0; j < NF_TW; ++j) b[j] = sP[c][ty + NF_SWH + r][tx + NF_SWH + j];
#pragma unroll
                for (int sft = 0; sft < 4; ++sft)
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        uint32_t v = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int j = 4 * d + k - sft;
                            if (j >= 0 && j < NF_TW) v |= b[j] << (8 * k);
                        }
                        T[c][sft][r][d] = v;
                    }
            }
        const uint32_t sqT = sB[(tx + NF_SWH) & 3][ty + NF_SWH][(tx + NF_SWH) >> 2];
        uint32_t est0 = 0, est1 = 0, wsum = 0;
#pragma unroll 1
        for (int dy = 0; dy < NF_SW; ++dy) {
            // candidate dwords of the 9 window rows: region dwords lx + q .. lx + q + 2, q = (w + dx) >> 2
            uint32_t Wd[CH][NF_TW][3];
            const uint32_t* rowp[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                rowp[c] = (const uint32_t*)&sP[c][ty + dy][0] + lx;
#pragma unroll
                for (int r = 0; r < NF_TW; ++r)
#pragma unroll
                    for (int d = 0; d < 3; ++d) Wd[c][r][d] = rowp[c][r * (NF2_PITCH / 4) + d];
            }
            // u = w + dx runs over [w, w + 30]; unrolled by four so that the shift u & 3 is a constant
#pragma unroll 1
            for (int q = 0; q < (3 + NF_SW + 3) / 4; ++q) {
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) {
                    const int dx = 4 * q + sft - w;           // wave-uniform
                    if (dx >= 0 && dx < NF_SW) {
                        uint32_t acc3[3] = {0, 0, 0};   // three independent chains
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int r = 0; r < NF_TW; ++r)
#pragma unroll
                                for (int d = 0; d < 3; ++d)
                                    acc3[d] = __builtin_amdgcn_udot4(Wd[c][r][d], T[c][sft][r][d], acc3[d], false);
                        const uint32_t acc = acc3[0] + acc3[1] + acc3[2];
                        const int cx = tx + dx;               // window position: class (cx & 3) == sft, index lx + q
                        const uint32_t sqW = sB[sft][ty + dy][lx + q];
                        const int dist = (int)(sqT + sqW - 2u * acc);
                        const int a = dist >> shift;
                        const uint32_t wgt = (uint32_t)(a < NF_WLDS ? s_w[a] : (a < nw ? wtab[a] : 0));
                        const int cy = ty + dy + NF_TWH;
                        est0 += wgt * sP[0][cy][cx + NF_TWH];
                        if (CH == 2) est1 += wgt * sP[CH - 1][cy][cx + NF_TWH];
                        wsum += wgt;
                    }
                }
                // the next four steps look one dword further right
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int r = 0; r < NF_TW; ++r) {
                        Wd[c][r][0] = Wd[c][r][1];
                        Wd[c][r][1] = Wd[c][r][2];
                        Wd[c][r][2] = rowp[c][r * (NF2_PITCH / 4) + q + 3];
                    }
            }
        }
        const int ox = blockIdx.x * NF2_TW + tx, oy = blockIdx.y * NF2_TH + ty;
        if (ox < W && oy < H) {
            uint8_t* out = dst + ((size_t)blockIdx.z * HW + (size_t)oy * W + ox) * CH;
            const uint32_t v0 = (est0 + wsum / 2) / wsum;
            out[0] = (uint8_t)(v0 > 255u ? 255u : v0);
            if (CH == 2) {
                const uint32_t v1 = (est1 + wsum / 2) / wsum;
                out[1] = (uint8_t)(v1 > 255u ? 255u : v1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The window-difference form for the two-channel (a, b) image with BLK vertically adjacent output pixels
// per thread: their windows at one search offset share all but one row each, so the per-row sums of
// squared differences R_i (BLK + 8 rows of 9 dwords) are computed once and the BLK distances are sliding
// sums of nine of them: (BLK + 8) * 9 / BLK difference pairs per pixel and candidate instead of 81.  (In
// the correlation form nothing is shared between neighbouring pixels: their own windows differ.)  A wave
// takes 64 consecutive columns (LDS reads of a row are consecutive dwords), wave w the rows BLK w ..
#define NF3_TW 64
template <int BLK>
__global__ __launch_bounds__(256) void k_nf_nlm_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                                     const int* __restrict__ wtab, int nw, int shift) {
    constexpr int TH = 4 * BLK;                       // tile rows
    constexpr int RH = TH + 2 * NF_BORDER, RW = NF3_TW + 2 * NF_BORDER;
    constexpr int PITCH = RW + 2;                     // dwords
    constexpr int NR = BLK + NF_TW - 1;               // window rows a thread works on
    __shared__ uint32_t sV[RH * PITCH];               // lane 0 = a, lane 1 = b of one pixel
    __shared__ int s_w[NF_WLDS];
    const int tid = threadIdx.x, tx = tid & 63, ty = (tid >> 6) * BLK;
    const size_t HW = (size_t)H * W;
    const uint8_t* img = src + (size_t)blockIdx.z * HW * 2;
    const int x0 = blockIdx.x * NF3_TW - NF_BORDER, y0 = blockIdx.y * TH - NF_BORDER;
    for (int e = tid; e < NF_WLDS; e += 256) s_w[e] = e < nw ? wtab[e] : 0;
    for (int e = tid; e < RH * RW; e += 256) {
        const int ry = e / RW, rx = e - ry * RW;
        const size_t q = (size_t)nf_reflect101(y0 + ry, H) * W + nf_reflect101(x0 + rx, W);
        sV[ry * PITCH + rx] = (uint32_t)img[2 * q] | ((uint32_t)img[2 * q + 1] << 16);
    }
    __syncthreads();
    nf_v2s T[NR][NF_TW];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < NF_TW; ++j) T[r][j] = nf_as_v2s(sV[(ty + NF_SWH + r) * PITCH + tx + NF_SWH + j]);
    uint32_t est0[BLK], est1[BLK], wsum[BLK];
#pragma unroll
    for (int o = 0; o < BLK; ++o) { est0[o] = 0; est1[o] = 0; wsum[o] = 0; }
    // distances from here on have weight zero (the table's non-zero prefix is nw entries of 2^shift each)
    const int zero_from = nw < (0x7fffffff >> shift) ? (nw << shift) : 0x7fffffff;
    nf_v2s Wn[NR][NF_TW];   // candidate rows: column slot (j + dx) % 9 holds window column j at step dx (see k_nf_nlm)
#pragma unroll 1
    for (int dy = 0; dy < NF_SW; ++dy) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int j = 0; j < NF_TW; ++j) Wn[r][j] = nf_as_v2s(sV[(ty + dy + r) * PITCH + tx + j]);
#pragma unroll 1
        for (int dx0 = 0; dx0 < NF_SW; dx0 += NF_TW) {
#pragma unroll
            for (int k = 0; k < NF_TW; ++k) {
                const int dx = dx0 + k;
                if (dx < NF_SW) {
                    int R[NR];
                    auto row_sum = [&](int r) {
                        int acc = 0;
#pragma unroll
                        for (int j = 0; j < NF_TW; ++j) {
                            const nf_v2s d = T[r][j] - Wn[r][(j + k) % NF_TW];
                            acc = __builtin_amdgcn_sdot2(d, d, acc, false);
                        }
                        R[r] = acc;
                    };
                    // Rows BLK-1 .. 8 belong to the windows of ALL of the thread's pixels: once their sum has
                    // reached the distance beyond which every weight is zero -- in every lane -- the candidate
                    // cannot contribute anywhere in the wave and the rest of its work is skipped (exact: only
                    // additions of zero are left out).  Most candidates of the chroma image end here after two rows.
                    int common = 0;
                    bool dead = false;
#pragma unroll
                    for (int r = BLK - 1; r < NF_TW; ++r) {
                        row_sum(r);
                        common += R[r];
                        if (r == BLK || r == NF_TW - 1) {
                            if (__all(common >= zero_from)) { dead = true; break; }
                        }
                    }
                    if (!dead) {
#pragma unroll
                        for (int r = 0; r < BLK - 1; ++r) row_sum(r);
#pragma unroll
                        for (int r = NF_TW; r < NR; ++r) row_sum(r);
                        int dist = common;
#pragma unroll
                        for (int r = 0; r < BLK - 1; ++r) dist += R[r];
#pragma unroll
                        for (int o = 0; o < BLK; ++o) {
                            if (o > 0) dist += R[o + NF_TW - 1] - R[o - 1];
                            const int a = dist >> shift;
                            const uint32_t wgt = (uint32_t)(a < NF_WLDS ? s_w[a] : (a < nw ? wtab[a] : 0));
                            const uint32_t v = sV[(ty + o + dy + NF_TWH) * PITCH + tx + dx + NF_TWH];
                            est0[o] += wgt * (v & 0xffffu);
                            est1[o] += wgt * (v >> 16);
                            wsum[o] += wgt;
                        }
                    }
                    if (dx + 1 < NF_SW) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) Wn[r][k] = nf_as_v2s(sV[(ty + dy + r) * PITCH + tx + dx + NF_TW]);
                    }
                }
            }
        }
    }
    const int ox = blockIdx.x * NF3_TW + tx;
#pragma unroll
    for (int o = 0; o < BLK; ++o) {
        const int oy = blockIdx.y * TH + ty + o;
        if (ox < W && oy < H) {
            uint8_t* out = dst + ((size_t)blockIdx.z * HW + (size_t)oy * W + ox) * 2;
            const uint32_t v0 = (est0[o] + wsum[o] / 2) / wsum[o], v1 = (est1[o] + wsum[o] / 2) / wsum[o];
            out[0] = (uint8_t)(v0 > 255u ? 255u : v0);
            out[1] = (uint8_t)(v1 > 255u ? 255u : v1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Correlation form with the row products shared down a column (the default).  The product sum of two
// windows is a sum of nine ROW products, and the row product of image row Y at column x with the row
// dy below and dx to the right belongs to the windows of nine vertically adjacent pixels: a thread that
// owns BLK consecutive pixels of one column computes BLK + 8 row products per candidate offset (three
// v_dot4_u32_u8 each and channel) and slides a nine-row sum down them -- 3 (BLK + 8) / BLK byte dot
// products per pixel, candidate and channel instead of 27.
// Alignment as in k_nf_nlm_corr (lanes take the columns x = 4 lx + wave, so a candidate's byte offset
// inside its dwords is the same for the whole wave), but the search offsets are walked by alignment
// class: for one class the pixel's own rows are shifted once, the candidate dwords of a search row
// slide on by one whole dword per step (eight steps, unrolled: no register moves), and the candidate's
// centre pixel is always a byte of its middle dword.
// Requires nw < NF_WLDS (the weight of a distance beyond the table's non-zero prefix is read from the
// zero padding).
#ifndef NF_VS_WPE
#define NF_VS_WPE(CH) ((CH) == 1 ? 3 : 2)
#endif
template <int CH, int BLK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NF_VS_WPE(CH), NF_VS_WPE(CH)))) void k_nf_nlm_vs(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                                   const int* __restrict__ wtab, int nw, int shift) {
    constexpr int TH = 2 * BLK;                        // tile rows: lanes 0..31 the upper BLK, 32..63 the lower
    constexpr int RH = TH + 2 * NF_BORDER;             // staged rows
    constexpr int BH = RH - NF_TW + 1;                 // rows of window positions
    constexpr int NR = BLK + NF_TW - 1;                // image rows under a thread's windows
    constexpr int PD = NF2_PITCH / 4;                  // staged row pitch in dwords
    __shared__ __attribute__((aligned(16))) uint8_t sP[CH][RH][NF2_PITCH];
    __shared__ uint32_t sB[4][RH][NF2_BP];             // box sums of squares, [column class][row][column / 4]
    __shared__ int s_w[NF_WLDS];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lx = lane & 31, ly = lane >> 5;
    const size_t HW = (size_t)H * W;
    const uint8_t* img = src + (size_t)blockIdx.z * HW * CH;
    const int x0 = blockIdx.x * NF2_TW - NF_BORDER, y0 = blockIdx.y * TH - NF_BORDER;
    for (int e = tid; e < NF_WLDS; e += 256) s_w[e] = e < nw ? wtab[e] : 0;
    for (int e = tid; e < RH * NF2_PITCH; e += 256) {
        const int ry = e / NF2_PITCH, rx = e - ry * NF2_PITCH;
        const size_t q = (size_t)nf_reflect101(y0 + ry, H) * W + nf_reflect101(x0 + imin(rx, NF2_RW - 1), W);
#pragma unroll
        for (int c = 0; c < CH; ++c) sP[c][ry][rx] = rx < NF2_RW ? img[CH * q + c] : (uint8_t)0;
    }
    __syncthreads();
    for (int e = tid; e < RH * NF2_BW; e += 256) {     // horizontal 9-sums of the squares
        const int ry = e / NF2_BW, rx = e - ry * NF2_BW;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < NF_TW; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const uint32_t v = sP[c][ry][rx + j];
                acc += v * v;
            }
        sB[rx & 3][ry][rx >> 2] = acc;
    }
    __syncthreads();
    if (tid < NF2_BW) {                                // vertical 9-sums in place, one column per thread
        uint32_t* col = &sB[tid & 3][0][tid >> 2];
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < NF_TW; ++i) run += col[i * NF2_BP];
        for (int ry = 0; ry < BH; ++ry) {
            const uint32_t top = col[ry * NF2_BP];
            col[ry * NF2_BP] = run;
            if (ry + 1 < BH) run += col[(ry + NF_TW) * NF2_BP] - top;
        }
    }
    __syncthreads();

    const int tx = 4 * lx + w;                         // this thread's column of the tile
    const int ty0 = ly * BLK;                          // ... and first row
    uint32_t sqT[BLK], wsum[BLK], est[CH][BLK];
#pragma unroll
    for (int o = 0; o < BLK; ++o) {
        sqT[o] = sB[(tx + NF_SWH) & 3][ty0 + o + NF_SWH][(tx + NF_SWH) >> 2];
        wsum[o] = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) est[c][o] = 0;
    }
    const int own_dw = lx + ((w + NF_SWH) >> 2);       // the own windows start at byte own_sh of this dword
    const int own_sh = 8 * ((w + NF_SWH) & 3);         // (wave-uniform, in bits)
#pragma unroll 1
    for (int cls = 0; cls < 4; ++cls) {
        // candidates whose first column is of class cls: dx = r4, r4 + 4, ...
        const int r4 = (cls - w) & 3, qd = (w + r4) >> 2, cbits = 8 * cls;
        // the own rows with their 9 bytes at bytes cls .. cls + 8 of three dwords, zero elsewhere
        uint32_t Ts[CH][NR][3];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const uint32_t* rp = (const uint32_t*)&sP[c][ty0 + NF_SWH + r][0] + own_dw;
                const unsigned long long d01 = (unsigned long long)rp[0] | ((unsigned long long)rp[1] << 32);
                const unsigned long long d12 = (unsigned long long)rp[1] | ((unsigned long long)rp[2] << 32);
                const uint32_t c0 = (uint32_t)(d01 >> own_sh), c1 = (uint32_t)(d12 >> own_sh), c2 = (rp[2] >> own_sh) & 0xffu;
                const unsigned long long lo = ((unsigned long long)c0 | ((unsigned long long)c1 << 32)) << cbits;
                const unsigned long long hi = ((unsigned long long)c1 | ((unsigned long long)c2 << 32)) << cbits;
                Ts[c][r][0] = (uint32_t)lo;
                Ts[c][r][1] = (uint32_t)(lo >> 32);
                Ts[c][r][2] = (uint32_t)(hi >> 32);
            }
#pragma unroll 1
        for (int dy = 0; dy < NF_SW; ++dy) {
            // candidate rows ty0 + dy + r; step k looks at dwords k .. k + 2 from here, kept in slots (k + d) % 3
            const uint32_t* rowp[CH];
            uint32_t Wd[CH][NR][3];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                rowp[c] = (const uint32_t*)&sP[c][ty0 + dy][0] + lx + qd;
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int d = 0; d < 3; ++d) Wd[c][r][d] = rowp[c][r * PD + d];
            }
            const uint32_t* sbp = &sB[cls][ty0 + dy][lx + qd];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (r4 + 4 * k < NF_SW) {              // wave-uniform
                    uint32_t RC[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        uint32_t acc = 0;
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d)
                                acc = __builtin_amdgcn_udot4(Wd[c][r][(k + d) % 3], Ts[c][r][d], acc, false);
                        RC[r] = acc;
                    }
                    uint32_t run = 0;
#pragma unroll
                    for (int r = 0; r < NF_TW; ++r) run += RC[r];
#pragma unroll
                    for (int o = 0; o < BLK; ++o) {
                        if (o > 0) run += RC[o + NF_TW - 1] - RC[o - 1];
                        const uint32_t sqW = sbp[o * NF2_BP + k];
                        const int dist = (int)(sqT[o] + sqW - 2u * run);
                        const int a = imin(dist >> shift, NF_WLDS - 1);
                        const uint32_t wgt = (uint32_t)s_w[a];
                        // the candidate's centre pixel: byte cls of the middle dword of its row 4
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                            est[c][o] += wgt * ((Wd[c][o + NF_TWH][(k + 1) % 3] >> cbits) & 0xffu);
                        wsum[o] += wgt;
                    }
                }
                // (steps are kept apart: scheduled across one another they need twice the registers)
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < 8) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int r = 0; r < NR; ++r) Wd[c][r][k % 3] = rowp[c][r * PD + k + 3];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const int ox = blockIdx.x * NF2_TW + tx;
#pragma unroll
    for (int o = 0; o < BLK; ++o) {
        const int oy = blockIdx.y * TH + ty0 + o;
        if (ox < W && oy < H) {
            uint8_t* out = dst + ((size_t)blockIdx.z * HW + (size_t)oy * W + ox) * CH;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const uint32_t v = (est[c][o] + wsum[o] / 2) / wsum[o];
                out[c] = (uint8_t)(v > 255u ? 255u : v);
            }
        }
    }
}

// OpenCV's integer Lab -> linear BGR (COLOR_Lab2LBGR, 8 bit).
__device__ __forceinline__ int nf_ab_to_xz(int v) {
    // C integer arithmetic: divisions truncate toward zero
    if (v <= 3390) return v * 108 / 841 - 290;       // 290 = 16384*16/116*108/841
    return v * v / 16384 * v / 16384;
}
__device__ __forceinline__ void nf_lab_to_bgr(const NfTables& t, const uint16_t* yf, int L, int a, int b, int out[3]) {
    const int y = yf[2 * L], ify = yf[2 * L + 1];
    const int adiv = ((5 * a * 53687 + (1 << 7)) >> 13) - 128 * 16384 / 500;
    const int bdiv = ((b * 41943 + (1 << 4)) >> 9) - 128 * 16384 / 200 - 1;
    const int x = nf_ab_to_xz(ify + adiv), z = nf_ab_to_xz(ify - bdiv);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v = (x * t.coef_inv[3 * c] + y * t.coef_inv[3 * c + 1] + z * t.coef_inv[3 * c + 2] + (1 << 13)) >> 14;
        v = nf_clampi(v, 0, 4095);
        out[c] = (v * 255) >> 12;
    }
}

// Lab -> BGR of the denoised planes, then (optionally) image - sat_u8(correlate(image,
// [[0,1,0],[1,-4,1],[0,1,0]])) with BORDER_REFLECT_101, saturating: a thread converts its pixel and
// its four neighbours.
__global__ __launch_bounds__(256) void k_nf_lab2bgr_sharpen(const uint8_t* __restrict__ lpl, const uint8_t* __restrict__ abpl,
                                                            uint8_t* __restrict__ bgr, int H, int W, int sharpen, NfTables t) {
    __shared__ uint16_t s_yf[512];
    for (int e = threadIdx.x; e < 512; e += 256) s_yf[e] = t.yf_tab[e];
    __syncthreads();
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const size_t HW = (size_t)H * W;
    const uint8_t* lp = lpl + (size_t)blockIdx.z * HW;
    const uint8_t* ab = abpl + (size_t)blockIdx.z * HW * 2;
    auto convert = [&](int yy, int xx, int out[3]) {
        const size_t q = (size_t)yy * W + xx;
        nf_lab_to_bgr(t, s_yf, lp[q], ab[2 * q], ab[2 * q + 1], out);
    };
    int c[3];
    convert(y, x, c);
    if (sharpen) {
        int n0[3], n1[3], n2[3], n3[3];
        convert(nf_reflect101(y - 1, H), x, n0);
        convert(nf_reflect101(y + 1, H), x, n1);
        convert(y, nf_reflect101(x - 1, W), n2);
        convert(y, nf_reflect101(x + 1, W), n3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int lap = nf_clampi(n0[k] + n1[k] + n2[k] + n3[k] - 4 * c[k], 0, 255);
            c[k] = nf_clampi(c[k] - lap, 0, 255);
        }
    }
    uint8_t* out = bgr + ((size_t)blockIdx.z * HW + (size_t)y * W + x) * 3;
    out[0] = (uint8_t)c[0]; out[1] = (uint8_t)c[1]; out[2] = (uint8_t)c[2];
}
