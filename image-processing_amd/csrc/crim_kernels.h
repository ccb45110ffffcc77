// HIP kernels of the Criminisi engine for gfx950 (wave64).  Every floating-point expression
// below is evaluated as written, in double precision, with FMA contraction disabled at build
// time (-ffp-contract=off): the confidence, data and priority terms and the search cost must
// reproduce NumPy's results bit for bit, because arg-max / arg-min ties decide which patch is
// filled next.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <math.h>

#include "crim_types.h"

// Diagnostic build only (-DCRIM_STAMPS): thread 0 of block 0 records the 100 MHz wall clock at
// named points of each iteration kernel into a buffer nothing else reads.
#ifdef CRIM_STAMPS
#define STAMP(slot)                                                                         \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)      \
            p.dbg[slot] = wall_clock64();                                                   \
    } while (0)
// the same for a block in the middle of a 1-D grid (steady state: warm instruction cache, loaded GPU)
#define STAMP_MID(slot)                                                                     \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 + 5)                            \
            p.dbg[slot] = wall_clock64();                                                   \
    } while (0)
// per-block records beyond the 64 named slots: entry (which = 0) and end (which = 1) of every block of a
// launch of at most 4096 blocks
#define STAMP_BLOCK(which)                                                                  \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 4096) p.dbg[64 + 2 * blockIdx.x + (which)] = wall_clock64(); \
        /* the shader clock beside the wall clock for one block: their ratio is the clock rate under load */ \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 + 5) p.dbg[56 + (which)] = clock64(); \
    } while (0)
// the same for thread 0 of the block with linear id `blk`
#define STAMP_AT(blk, slot)                                                                 \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == (blk)) p.dbg[slot] = wall_clock64();          \
    } while (0)
// the same under a condition (a tile in the middle of a block's walk)
#define TSTAMP(cond, slot) do { if ((cond) && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 + 5) p.dbg[slot] = wall_clock64(); } while (0)
#else
#define TSTAMP(cond, slot) do {} while (0)
#define STAMP_AT(blk, slot) do {} while (0)
#define STAMP(slot) do {} while (0)
#define STAMP_MID(slot) do {} while (0)
#define STAMP_BLOCK(which) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// BORDER_REFLECT_101 for offsets of one pixel (cv2.Laplacian / cv2.filter2D defaults,
// criminisi_inpainter.py:103,149-150).
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// Read-only, wave-uniform data written by an EARLIER kernel (the target descriptor) is read
// through the constant address space so that the compiler emits scalar loads and the values
// become SGPR operands of the dot products.
typedef const uint32_t __attribute__((address_space(4))) * const_u32_ptr;
__device__ __forceinline__ const_u32_ptr as_const_u32(const void* p) {
    return (const_u32_ptr)(uintptr_t)p;
}

// Workgroups are handed to the 8 XCDs round-robin by linear id.  A 1-D grid of `per_image` blocks
// for each of `nimg` images is decoded so that every block of image i runs on XCD i % 8 (its L2
// then serves what the blocks of one image share: overlapping tile rows, halos, descriptors).
// Falls back to image-major order when the image count is not a multiple of 8.
__device__ __forceinline__ void xcd_decode(unsigned id, int per_image, int nimg, int* img, int* sub) {
    if ((nimg & 7) == 0) {
        const unsigned r = id & 7u, q = id >> 3;
        *sub = (int)(q % (unsigned)per_image);
        *img = (int)(8u * (q / (unsigned)per_image) + r);
    } else {
        *sub = (int)(id % (unsigned)per_image);
        *img = (int)(id / (unsigned)per_image);
    }
}

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Cross-lane helpers on the DPP path (no LDS round trip, unlike ds_bpermute-based shuffles).
// Lane l receives lane l ^ 1 / l ^ 2 of its quad:
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp_i32<CTRL>(__double2loint(v)), hi = dpp_i32<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
#define DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
// Maximum over the 64 lanes of a wave, returned to every lane: quads, half rows, rows by DPP
// permutes, then the row results are carried along by row broadcasts into lane 63.
__device__ __forceinline__ int wave_max_i32(int v) {
    v = imax_(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = imax_(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = imax_(v, dpp_i32<0x141>(v));   // row_half_mirror
    v = imax_(v, dpp_i32<0x140>(v));   // row_mirror: every lane holds its row's maximum
    v = imax_(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast15 into rows 1, 3
    v = imax_(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast31 into rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// The same ladder for a (key, index) pair under a strict total order `better(ka, ia, kb, ib)` (is a
// better than b): arg-max of the priority stage, arg-min of the search.  Every lane receives the
// wave's best pair.  Six DPP steps of three moves each, against six ds_bpermute round trips of a
// shuffle-based reduction.
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_rows_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false);   // lanes of other rows keep v
}
template <typename Better>
__device__ __forceinline__ void wave_best_pair(double& k, int& i, Better better) {
#define CRIM_PAIR_STEP(CTRL, ROWS)                                                                   \
    {                                                                                                \
        const double ok = __hiloint2double(dpp_rows_i32<CTRL, ROWS>(__double2hiint(k)),              \
                                           dpp_rows_i32<CTRL, ROWS>(__double2loint(k)));             \
        const int oi = dpp_rows_i32<CTRL, ROWS>(i);                                                  \
        if (better(ok, oi, k, i)) { k = ok; i = oi; }                                                \
    }
    CRIM_PAIR_STEP(DPP_QUAD_XOR1, 0xf)
    CRIM_PAIR_STEP(DPP_QUAD_XOR2, 0xf)
    CRIM_PAIR_STEP(0x141, 0xf)   // row_half_mirror
    CRIM_PAIR_STEP(0x140, 0xf)   // row_mirror: every lane holds its row's best
    CRIM_PAIR_STEP(0x142, 0xa)   // row_bcast15 into rows 1, 3
    CRIM_PAIR_STEP(0x143, 0xc)   // row_bcast31 into rows 2, 3: lane 63 holds the wave's best
#undef CRIM_PAIR_STEP
    k = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(k), 63), __builtin_amdgcn_readlane(__double2loint(k), 63));
    i = __builtin_amdgcn_readlane(i, 63);
}
// Sum / bitwise OR over the 64 lanes of a wave, returned to every lane (lanes of rows a broadcast
// step does not reach contribute 0 there).
template <bool OR>
__device__ __forceinline__ int wave_fold_i32(int v) {
    auto f = [](int a, int b) { return OR ? (a | b) : (a + b); };
    v = f(v, __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR1, 0xf, 0xf, false));
    v = f(v, __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR2, 0xf, 0xf, false));
    v = f(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
    v = f(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
    v = f(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = f(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// A rectangular window of a 2-D field: element (y, x) in image coordinates lives `stride` bytes
// apart at base + ((y - oy) * pitch + (x - ox)) * stride.  The same code reads whole frames in
// global memory (oy = ox = 0, pitch = W, stride = sizeof(T)) and the interleaved records of a tile
// staged in LDS.
template <typename T>
struct View {
    const unsigned char* base;
    int pitch, oy, ox, stride;
    __device__ __forceinline__ T at(int y, int x) const {
        return *(const T*)(base + (size_t)((y - oy) * pitch + (x - ox)) * stride);
    }
};

// Reads element e of a row-major ph x pw window of the confidence map (flattened order).
struct PatchReader {
    View<double> c;
    int r0, c0, pw;
    unsigned inv;  // ceil(65536 / pw): exact floor(e / pw) for e < 2048, pw <= 31
    __device__ __forceinline__ double operator()(int e) const {
        const unsigned q = ((unsigned)e * inv) >> 16;
        const int col = e - (int)q * pw;
        return c.at(r0 + (int)q, c0 + col);
    }
};

// 3x3 mask neighbourhood of pixel (y, x), row-major, from a whole-frame mask with
// BORDER_REFLECT_101 (the default border of cv2.Laplacian / cv2.filter2D).
__device__ __forceinline__ void mask_nb_global(const uint8_t* mask, int H, int W, int y, int x,
                                               int nb[9]) {
    const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
    const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
    const uint8_t* r0 = mask + (size_t)ym * W;
    const uint8_t* r1 = mask + (size_t)y * W;
    const uint8_t* r2 = mask + (size_t)yp * W;
    nb[0] = r0[xm]; nb[1] = r0[x]; nb[2] = r0[xp];
    nb[3] = r1[xm]; nb[4] = r1[x]; nb[5] = r1[xp];
    nb[6] = r2[xm]; nb[7] = r2[x]; nb[8] = r2[xp];
}

// Fill-front weight from the neighbourhood: sat_u8(Laplacian_ksize3(mask)) with the aperture
// [[2,0,2],[0,-8,0],[2,0,2]] (criminisi_inpainter.py:103): positive only on a known pixel with a
// masked diagonal neighbour.
__device__ __forceinline__ int front_from_nb(const int nb[9]) {
    const int v = 2 * (nb[0] + nb[2] + nb[6] + nb[8]) - 8 * nb[4];
    return v < 0 ? 0 : v;
}

__device__ __forceinline__ int front_weight(const uint8_t* mask, int H, int W, int y, int x,
                                            int nb[9]) {
    mask_nb_global(mask, H, W, y, x, nb);
    return front_from_nb(nb);
}

struct FrontEval {
    double cnew, data, prio;
};

#ifndef PG
#define PG 4   // lanes that cooperate on one front pixel (1, 2, 4 or 8)
#endif

// NumPy's pairwise summation (np.sum in criminisi_inpainter.py:124-125 flattens the patch in C
// order) of n <= 128 doubles, computed by PG cooperating lanes.  Fewer than 8 elements are added
// sequentially.  Otherwise NumPy keeps eight accumulators r[0..7] (r[j] sums elements j, j+8,
// ...), combines them as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and adds the n%8 tail.  Lane l owns
// the 8/PG adjacent accumulators r[l*8/PG ...], reduces them in that tree order, and the lanes
// finish the tree with an xor butterfly (floating-point addition is commutative, so both
// partners of an exchange obtain identical bits).  Every lane returns the full sum.
__device__ __forceinline__ double pw_leaf8(const PatchReader& a, int lo, int n, int l) {
    if (n < 8) {
        double r = -0.0;
        for (int i = 0; i < n; ++i) r += a(lo + i);
        return r;
    }
    constexpr int A = 8 / PG;
    const int lim = n - (n % 8);
    double r[A];
#pragma unroll
    for (int k = 0; k < A; ++k) r[k] = a(lo + l * A + k);
    for (int i = 8; i < lim; i += 8) {
#pragma unroll
        for (int k = 0; k < A; ++k) r[k] += a(lo + i + l * A + k);
    }
    double t;
    if (A == 8) t = ((r[0] + r[1 % A]) + (r[2 % A] + r[3 % A])) + ((r[4 % A] + r[5 % A]) + (r[6 % A] + r[7 % A]));
    else if (A == 4) t = (r[0] + r[1 % A]) + (r[2 % A] + r[3 % A]);
    else if (A == 2) t = r[0] + r[1 % A];
    else t = r[0];
#pragma unroll
    for (int off = 1; off < PG; off <<= 1) t = t + __shfl_xor(t, off, PG);
    for (int i = lim; i < n; ++i) t += a(lo + i);
    return t;
}
// Larger windows are halved (first half rounded down to a multiple of 8) and the halves summed,
// as NumPy does.  Depth 3 covers n <= 1024, i.e. patch edges up to 31.
template <int D>
__device__ __forceinline__ double pw_sum8(const PatchReader& a, int lo, int n, int l) {
    if (n <= 128) return pw_leaf8(a, lo, n, l);
    int n2 = n / 2;
    n2 -= n2 % 8;
    const double x = pw_sum8<D - 1>(a, lo, n2, l);
    const double y = pw_sum8<D - 1>(a, lo + n2, n - n2, l);
    return x + y;
}
template <>
__device__ __forceinline__ double pw_sum8<0>(const PatchReader& a, int lo, int n, int l) {
    return pw_leaf8(a, lo, n, l);
}

// Measured alternatives that were slower and are not kept: caching the gradient arg-max per
// pixel behind dirty flags (42 us vs 35 us per launch of 128 images: the extra global reads cost
// more than the scan saves), four tiles per 256-thread block (39 us), register caps through
// launch bounds (spills).

// Confidence (criminisi_inpainter.py:114-127), data term (:129-214) and priority (:112) of one
// front pixel, computed by PG cooperating lanes (l = lane index within the group; all PG lanes
// must call this together and all receive the result).  conf is the confidence map before this
// iteration's update, nb the pixel's 3x3 mask neighbourhood, F its front weight.
// D is the halving depth of the pairwise sum: 0 when the patch is known to hold at most 128 pixels.
// New confidence of a front pixel alone (criminisi_inpainter.py:114-127): the patch clipped to the
// image, summed in NumPy's order by PG cooperating lanes, over patch_size squared.
template <int D = 3>
__device__ __forceinline__ double front_confidence(const Params& p, const View<double>& conf, int y, int x,
                                                   int l) {
    const int H = p.H, W = p.W, half = p.half;
    const int r0 = imax(0, y - half), r1 = imin(y + half, H - 1);
    const int c0 = imax(0, x - half), c1 = imin(x + half, W - 1);
    const int ph = r1 - r0 + 1, pw = c1 - c0 + 1;
    PatchReader rd;
    rd.c = conf; rd.r0 = r0; rd.c0 = c0; rd.pw = pw;
    rd.inv = (65536u + (unsigned)pw - 1u) / (unsigned)pw;
    return pw_sum8<D>(rd, 0, ph * pw, l) / (double)p.psq;
}

template <int D = 3>
__device__ __forceinline__ FrontEval eval_front_pixel(const Params& p, const View<int>& grad,
                                                      const View<double>& conf, int y, int x, int F,
                                                      const int nb[9], int l) {
    const int H = p.H, W = p.W, half = p.half;
    const int r0 = imax(0, y - half), r1 = imin(y + half, H - 1);
    const int c0 = imax(0, x - half), c1 = imin(x + half, W - 1);
    const int ph = r1 - r0 + 1, pw = c1 - c0 + 1;
    const int n = ph * pw;
    PatchReader rd;
    rd.c = conf; rd.r0 = r0; rd.c0 = c0; rd.pw = pw;
    rd.inv = (65536u + (unsigned)pw - 1u) / (unsigned)pw;
    FrontEval out;
    out.cnew = pw_sum8<D>(rd, 0, n, l) / (double)p.psq;

    // mask normal (:138-164): Sobel/4 correlation, then normalised
    const double a = nb[0], b = nb[1], c = nb[2], d = nb[3], f = nb[5], g = nb[6], h = nb[7],
                 k = nb[8];
    const double nx = .25 * a - .25 * c + .5 * d - .5 * f + .25 * g - .25 * k;
    const double ny = -.25 * a - .5 * b - .25 * c + .25 * g + .5 * h + .25 * k;
    double nrm = sqrt(ny * ny + nx * nx);
    if (nrm == 0) nrm = 1;
    const double ux = nx / nrm, uy = ny / nrm;

    // first maximum of the gradient magnitude inside the patch (:183-211).  grad holds
    // (2*gy, 2*gx) as integers; comparing 4*|g|^2 orders exactly like comparing sqrt(|g|^2).
    // Lane l scans elements l, l+PG, ... of the flattened patch; the group keeps the largest
    // magnitude and, among equals, the smallest flattened index.
    int best = -1, beste = 0x7fffffff, bpk = 0;
    for (int e = l; e < n; e += PG) {
        const unsigned q = ((unsigned)e * rd.inv) >> 16;
        const int col = e - (int)q * pw;
        const int gpk = grad.at(r0 + (int)q, c0 + col);
        const int gy2 = (int)(short)(gpk & 0xffff), gx2 = gpk >> 16;
        const int m = gy2 * gy2 + gx2 * gx2;
        if (m > best) { best = m; beste = e; bpk = gpk; }
    }
#pragma unroll
    for (int off = 1; off < PG; off <<= 1) {
        const int om = __shfl_xor(best, off, PG);
        const int oe = __shfl_xor(beste, off, PG);
        const int ok = __shfl_xor(bpk, off, PG);
        if (om > best || (om == best && oe < beste)) { best = om; beste = oe; bpk = ok; }
    }
    const int by = (int)(short)(bpk & 0xffff), bx = bpk >> 16;
    const double Gy = (double)by * 0.5, Gx = (double)bx * 0.5;
    const double t0 = ux * Gy, t1 = uy * Gx;
    out.data = sqrt(t0 * t0 + t1 * t1) + 0.001;
    out.prio = out.cnew * out.data * (double)F;
    return out;
}

// 2*d/dy and 2*d/dx of np.gradient (criminisi_inpainter.py:174) on the grey image with masked
// pixels read as 0 (:171), packed as two int16.
__device__ __forceinline__ int grad_packed(const uint8_t* grey, const uint8_t* mask, int H, int W,
                                           int y, int x) {
    auto G = [&](int yy, int xx) -> int {
        const size_t i = (size_t)yy * W + xx;
        return mask[i] ? 0 : (int)grey[i];
    };
    int gy2, gx2;
    if (H == 1) gy2 = 0;
    else if (y == 0) gy2 = 2 * (G(1, x) - G(0, x));
    else if (y == H - 1) gy2 = 2 * (G(y, x) - G(y - 1, x));
    else gy2 = G(y + 1, x) - G(y - 1, x);
    if (W == 1) gx2 = 0;
    else if (x == 0) gx2 = 2 * (G(y, 1) - G(y, 0));
    else if (x == W - 1) gx2 = 2 * (G(y, x) - G(y, x - 1));
    else gx2 = G(y, x + 1) - G(y, x - 1);
    return (gy2 & 0xffff) | (int)((unsigned)gx2 << 16);
}

// Integer BGR -> Lab (cv2.COLOR_BGR2LAB on uint8, criminisi_inpainter.py:225) and the five
// search planes derived from it: Lab - 128 per channel and Q = sum (Lab-128)^2 split as
// Q % 255 and Q / 255, so that sum k*Q = dot(Q%255, k) + dot(Q/255, 255*k) stays in byte dot
// products.
// (gamma_tab, cbrt_tab: the two look-up tables, wherever the caller keeps them)
__device__ __forceinline__ void lab_planes(const Params& p, const uint16_t* gamma_tab, const uint16_t* cbrt_tab,
                                           int B_, int G_, int R_, uint8_t out[CRIM_NPL]) {
    const int b = gamma_tab[B_], g = gamma_tab[G_], r = gamma_tab[R_];
    const int fX = cbrt_tab[(b * p.coef[0] + g * p.coef[1] + r * p.coef[2] + 2048) >> 12];
    const int fY = cbrt_tab[(b * p.coef[3] + g * p.coef[4] + r * p.coef[5] + 2048) >> 12];
    const int fZ = cbrt_tab[(b * p.coef[6] + g * p.coef[7] + r * p.coef[8] + 2048) >> 12];
    int L = (296 * fY - 1336934 + 16384) >> 15;
    int a = (500 * (fX - fY) + 128 * 32768 + 16384) >> 15;
    int bb = (200 * (fY - fZ) + 128 * 32768 + 16384) >> 15;
    L = imin(imax(L, 0), 255) - 128;
    a = imin(imax(a, 0), 255) - 128;
    bb = imin(imax(bb, 0), 255) - 128;
    const int Q = L * L + a * a + bb * bb;
    out[0] = (uint8_t)(int8_t)L;
    out[1] = (uint8_t)(int8_t)a;
    out[2] = (uint8_t)(int8_t)bb;
    out[3] = (uint8_t)(Q % 255);
    out[4] = (uint8_t)(Q / 255);
    // Q <= 3 * 128^2 = 49152; 24576 - Q = 256*h + l with l in [-128,127], h in [-96,96]: both
    // fit the signed bytes the i8 matrix cores take.  (The negated quantity is stored so that the
    // matrix-core search accumulates -S with nothing but left shifts between its passes.)
    const int Qc = 24576 - Q;
    const int h = (Qc + 128) >> 8;
    out[5] = (uint8_t)(int8_t)(Qc - 256 * h);
    out[6] = (uint8_t)(int8_t)h;
}

// ------------------------------------------------------------------------------------------
// upload-time kernels
// ------------------------------------------------------------------------------------------

// Mask recipe of the reference's caller (src/utils.py:79-100) on the device: grey <= threshold
// (cv2.inRange(grey, 0, 10)), a border_width-pixel frame cleared, one k x k box dilation
// (cv2.dilate anchors an even box at k/2: offsets -k/2 .. k-1-k/2; pixels beyond the image never
// win), then {0,1}.  Reads prm.bgr, writes prm.mask.
__global__ __launch_bounds__(256) void k_build_mask(Params p, int first, int threshold, int border,
                                                    int k) {
    const int b = first + blockIdx.z;
    const int y = blockIdx.y * CRIM_TILE + (threadIdx.x >> 4);
    const int x = blockIdx.x * CRIM_TILE + (threadIdx.x & 15);
    if (y >= p.H || x >= p.W) return;
    const size_t HW = (size_t)p.H * p.W;
    const uint8_t* img = p.bgr + b * HW * 3;
    const int a = k / 2;
    int m = 0;
    for (int dy = -a; dy < k - a && !m; ++dy)
        for (int dx = -a; dx < k - a; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < border || yy >= p.H - border || xx < border || xx >= p.W - border) continue;
            const uint8_t* px = img + ((size_t)yy * p.W + xx) * 3;
            const int g = (px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + 16384) >> 15;
            if (g <= threshold) { m = 1; break; }
        }
    p.mask[b * HW + (size_t)y * p.W + x] = (uint8_t)m;
}

// Normalises the mask to {0,1}, derives confidence, grey and the search planes and counts the
// masked pixels.  Every array but the planes is a plain [H*W] vector, so a thread takes four
// consecutive pixels of the flattened image (1024 per block): dword mask / grey / plane accesses
// and 32-byte confidence stores instead of single bytes.  Four pixels stay in one row, and their
// plane bytes in one aligned dword, when W is a multiple of 4; other widths go pixel by pixel.
// src_bgr / src_mask: the uploaded images where they are not in the engine's buffers yet (device-to-device
// uploads: images first .. of the call, dense) -- the kernel then fills prm.bgr / prm.mask itself instead of
// a copy made beforehand; null: the buffers already hold them.
__global__ __launch_bounds__(256) void k_init_pixels(Params p, int first, const uint8_t* __restrict__ src_bgr,
                                                     const uint8_t* __restrict__ src_mask) {
    const int b = first + blockIdx.y;
    const size_t HW = (size_t)p.H * p.W;
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    uint8_t* mk = p.mask + b * HW;
    uint8_t* bgr_out = p.bgr + b * HW * 3;
    const uint8_t* bgr = src_bgr ? src_bgr + (size_t)blockIdx.y * HW * 3 : bgr_out;
    const uint8_t* mk_in = src_mask ? src_mask + (size_t)blockIdx.y * HW : mk;
    uint8_t* grey = p.grey + b * HW;
    double* c0 = p.conf + (b * 2 + 0) * HW;
    double* c1 = p.conf + (b * 2 + 1) * HW;
    uint8_t* planes = p.planes + (size_t)b * CRIM_NPL * p.H * p.Wp;
    const size_t PS = (size_t)p.H * p.Wp;
    int cnt = 0, chroma = 0;
    // the two tables of the Lab conversion in LDS: 24 look-ups per thread, scattered over 6.5 KB
    __shared__ uint16_t s_gamma[CRIM_GAMMA_N], s_cbrt[CRIM_CBRT_N];
    for (int e = threadIdx.x; e < CRIM_GAMMA_N / 2; e += 256) ((uint32_t*)s_gamma)[e] = ((const uint32_t*)p.gamma_tab)[e];
    for (int e = threadIdx.x; e < CRIM_CBRT_N / 2; e += 256) ((uint32_t*)s_cbrt)[e] = ((const uint32_t*)p.cbrt_tab)[e];
    __syncthreads();
    auto one = [&](int B_, int G_, int R_, uint8_t pl[CRIM_NPL]) -> uint8_t {
        lab_planes(p, s_gamma, s_cbrt, B_, G_, R_, pl);
        chroma |= (pl[1] | pl[2]) ? 1 : 0;
        return (uint8_t)((B_ * 3735 + G_ * 19235 + R_ * 9798 + 16384) >> 15);
    };
    if ((p.W & 3) == 0 && i0 + 4 <= HW) {
        const uint32_t m4 = *(const uint32_t*)(mk_in + i0);
        const uint32_t q0 = *(const uint32_t*)(bgr + 3 * i0), q1 = *(const uint32_t*)(bgr + 3 * i0 + 4),
                       q2 = *(const uint32_t*)(bgr + 3 * i0 + 8);
        if (src_bgr) {
            *(uint32_t*)(bgr_out + 3 * i0) = q0; *(uint32_t*)(bgr_out + 3 * i0 + 4) = q1; *(uint32_t*)(bgr_out + 3 * i0 + 8) = q2;
        }
        const uint8_t px[12] = {(uint8_t)q0, (uint8_t)(q0 >> 8), (uint8_t)(q0 >> 16), (uint8_t)(q0 >> 24),
                                (uint8_t)q1, (uint8_t)(q1 >> 8), (uint8_t)(q1 >> 16), (uint8_t)(q1 >> 24),
                                (uint8_t)q2, (uint8_t)(q2 >> 8), (uint8_t)(q2 >> 16), (uint8_t)(q2 >> 24)};
        uint32_t mo = 0, go = 0, po[CRIM_NPL] = {0, 0, 0, 0, 0, 0, 0};
        double cv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = ((m4 >> (8 * j)) & 0xffu) ? 1 : 0;
            cnt += m;
            mo |= (uint32_t)m << (8 * j);
            cv[j] = 1.0 - (double)m;
            uint8_t pl[CRIM_NPL];
            go |= (uint32_t)one(px[3 * j], px[3 * j + 1], px[3 * j + 2], pl) << (8 * j);
#pragma unroll
            for (int k2 = 0; k2 < CRIM_NPL; ++k2) po[k2] |= (uint32_t)pl[k2] << (8 * j);
        }
        *(uint32_t*)(mk + i0) = mo;
        *(uint32_t*)(grey + i0) = go;
        *(double2*)(c0 + i0) = make_double2(cv[0], cv[1]);
        *(double2*)(c0 + i0 + 2) = make_double2(cv[2], cv[3]);
        *(double2*)(c1 + i0) = make_double2(cv[0], cv[1]);
        *(double2*)(c1 + i0 + 2) = make_double2(cv[2], cv[3]);
        const size_t y = i0 / p.W, x = i0 - y * p.W;
#pragma unroll
        for (int k2 = 0; k2 < CRIM_NPL; ++k2) *(uint32_t*)(planes + k2 * PS + y * p.Wp + x) = po[k2];
    } else {
        for (int j = 0; j < 4; ++j) {
            const size_t i = i0 + j;
            if (i >= HW) break;
            const int m = mk_in[i] ? 1 : 0;
            cnt += m;
            mk[i] = (uint8_t)m;
            const double c = 1.0 - (double)m;
            c0[i] = c;
            c1[i] = c;
            uint8_t pl[CRIM_NPL];
            const uint8_t cb = bgr[3 * i], cg = bgr[3 * i + 1], cr = bgr[3 * i + 2];
            if (src_bgr) { bgr_out[3 * i] = cb; bgr_out[3 * i + 1] = cg; bgr_out[3 * i + 2] = cr; }
            grey[i] = one(cb, cg, cr, pl);
            const size_t y = i / p.W, x = i - y * p.W;
            for (int k2 = 0; k2 < CRIM_NPL; ++k2) planes[k2 * PS + y * p.Wp + x] = pl[k2];
        }
    }
    // block totals: wave sums by a shuffle tree, one atomic per block
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    __shared__ int s_cnt[4];
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    const int any_chroma = __syncthreads_or(chroma);
    if (threadIdx.x == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (total) atomicAdd(&p.st[b].remaining, total);
        if (any_chroma) atomicOr(&p.st[b].chroma, 1);
    }
}

// Gradient field, window-validity map and the per-tile count of front pixels; needs the
// normalised mask of the whole image.  One block per strip of 16 rows x 256 columns (a row of 16 tiles);
// a thread owns one column of the strip and walks down its 16 rows, so every row is read and written
// as 256 consecutive elements.
#define IF_W 256
__global__ __launch_bounds__(256) void k_init_fields(Params p, int first) {
    const int b = first + blockIdx.z;
    const int x0 = blockIdx.x * IF_W, y0 = blockIdx.y * CRIM_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W, P = p.P;
    const size_t HW = (size_t)H * W;
    const uint8_t* mk = p.mask + b * HW;
    const uint8_t* gr = p.grey + b * HW;
    // Mask rows of the strip grown by P-1 to the right and below as bit rows (bit j of word k = pixel
    // x0 + 64 k + j masked or beyond the image), by wave ballots; a window is inadmissible when the OR of
    // its P rows has a bit set among its P columns.
    constexpr int NRM = CRIM_TILE + CRIM_PMAX - 1;
    __shared__ unsigned long long s_bits[NRM][5];
    // the strip grown by one pixel: the mask with BORDER_REFLECT_101 (front weight), and the grey
    // values with masked pixels read as 0 (gradient; positions beyond the image are never used)
    constexpr int TSW = IF_W + 2, TSP = IF_W + 8;
    __shared__ uint8_t s_mk[(CRIM_TILE + 2) * TSP], s_gg[(CRIM_TILE + 2) * TSP];
    static_assert(TSW <= 512, "two columns per thread at most");
    {
        // (this thread's columns: tid, and for the first TSW - 256 threads tid + 256)
        const int xx0 = imin(imax(x0 - 1 + tid, -1), W), xx1 = imin(imax(x0 - 1 + tid + 256, -1), W);
        const int rx0 = reflect101(xx0, W), rx1 = reflect101(xx1, W);
        const int cx0 = imin(imax(xx0, 0), W - 1), cx1 = imin(imax(xx1, 0), W - 1);
#pragma unroll 6
        for (int i = 0; i < CRIM_TILE + 2; ++i) {
            const int yy = imin(imax(y0 - 1 + i, -1), H);
            const size_t rrow = (size_t)reflect101(yy, H) * W, crow = (size_t)imin(imax(yy, 0), H - 1) * W;
            s_mk[i * TSP + tid] = mk[rrow + rx0];
            s_gg[i * TSP + tid] = mk[crow + cx0] ? (uint8_t)0 : gr[crow + cx0];
            if (tid < TSW - 256) {
                s_mk[i * TSP + tid + 256] = mk[rrow + rx1];
                s_gg[i * TSP + tid + 256] = mk[crow + cx1] ? (uint8_t)0 : gr[crow + cx1];
            }
        }
    }
    const int nrows = CRIM_TILE + P - 1;
    for (int r0 = 0; r0 < nrows; r0 += 8) {   // eight rows' loads travel together
        const int xa = x0 + tid, xb = x0 + IF_W + lane;
        bool m[8], m2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int yy = y0 + r0 + k;
            const bool out = yy >= H || r0 + k >= nrows;
            m[k] = (out || xa >= W) ? true : mk[(size_t)yy * W + xa] != 0;
            m2[k] = true;
            if (wave == 0) m2[k] = (out || xb >= W) ? true : mk[(size_t)yy * W + xb] != 0;   // right of the strip
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long bal = __ballot(m[k]), bal2 = __ballot(m2[k]);
            if (lane == 0 && r0 + k < nrows) {
                s_bits[r0 + k][wave] = bal;
                if (wave == 0) s_bits[r0 + k][4] = bal2;
            }
        }
    }
    __syncthreads();
    const int x = x0 + tid;
    const unsigned long long pm = (1ull << P) - 1ull;
    // bit r: the P columns from x of mask row y0 + r hold a masked pixel
    unsigned long long hbits = 0;
    for (int r = 0; r < nrows; ++r) {
        unsigned long long lo = s_bits[r][wave] >> lane;
        if (lane) lo |= s_bits[r][wave + 1] << (64 - lane);
        hbits |= (unsigned long long)((lo & pm) != 0) << r;
    }
    int cnt = 0;
    for (int py = 0; py < CRIM_TILE; ++py) {
        const int y = y0 + py;
        int inv_bit = 1;
        if (y < H && x < W) {
            {
                // np.gradient (:174): central differences, one-sided at the image border (grad_packed)
                const uint8_t* g0 = s_gg + (py + 1) * TSP + (tid + 1);
                int gy2, gx2;
                if (H == 1) gy2 = 0;
                else if (y == 0) gy2 = 2 * ((int)g0[TSP] - (int)g0[0]);
                else if (y == H - 1) gy2 = 2 * ((int)g0[0] - (int)g0[-TSP]);
                else gy2 = (int)g0[TSP] - (int)g0[-TSP];
                if (W == 1) gx2 = 0;
                else if (x == 0) gx2 = 2 * ((int)g0[1] - (int)g0[0]);
                else if (x == W - 1) gx2 = 2 * ((int)g0[0] - (int)g0[-1]);
                else gx2 = (int)g0[1] - (int)g0[-1];
                p.grad[b * HW + (size_t)y * W + x] = (gy2 & 0xffff) | (int)((unsigned)gx2 << 16);
            }
            const int inv = ((hbits >> py) & pm) != 0;
            p.invalid[((size_t)b * H + y) * p.Wi + x] = (uint8_t)inv;
            inv_bit = inv;
            const uint8_t* m0 = s_mk + py * TSP + tid;   // top-left of the 3x3 neighbourhood
            cnt += (m0[TSP + 1] == 0 && (m0[0] + m0[2] + m0[2 * TSP] + m0[2 * TSP + 2]) > 0) ? 1 : 0;
        }
        // bitmap form: a wave covers 64 origins of the row; origins beyond the image stay 1
        const unsigned long long bal = __ballot(inv_bit);
        const int word = (x0 >> 6) + wave;
        if (lane == 0 && y < H && word < p.Wb) p.invbits[((size_t)b * H + y) * p.Wb + word] = bal;
    }
    // front pixels per 16 x 16 tile: the 16 columns of a tile are 16 neighbouring lanes
    cnt += __shfl_xor(cnt, 1);
    cnt += __shfl_xor(cnt, 2);
    cnt += __shfl_xor(cnt, 4);
    cnt += __shfl_xor(cnt, 8);
    const int tX = (x0 >> 4) + (tid >> 4);
    if ((tid & 15) == 0 && tX < p.tnx) {
        const int tile = blockIdx.y * p.tnx + tX;
        p.tfront[(size_t)b * p.tnx * p.tny + tile] = cnt;
        if (cnt) p.tlist[(size_t)b * 2 * p.tcap + atomicAdd(&p.st[b].tcount, 1u)] = tile;   // buffer 0 (tlsel = 0)
    }
}

__global__ void k_init_state(Params p, int first, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    ImgState* s = p.st + first + i;
    s->iter = 0; s->live = 1; s->cur = 0; s->remaining = 0;
    s->tcount = 0; s->chroma = 0;
    s->pcount = 0; s->scount = 0; s->stalled = 0; s->bound = 0x7fffffff;
    s->rebuilds = 0; s->tlsel = 0;
    p.desc[first + i].live = 0;
}

// ------------------------------------------------------------------------------------------
// iteration kernel 1: front, confidence, data, priority; per-tile arg-max
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ bool better_prio(double pa, int ia, double pb, int ib) {
    return pa > pb || (pa == pb && ia < ib);
}

// The priority stage works on 16x16 pixel tiles that hold at least one front pixel.  tfront
// counts the front pixels of every tile exactly (maintained by k_update); tlist names the tiles
// to visit.  A tile is appended when its count leaves zero and stays listed, possibly with a zero
// count or twice, until the next rebuild; visiting such an entry is harmless.

// Rebuilds the tile list from the counts (one block of 1024 threads per image).
__global__ __launch_bounds__(1024) void k_tiles(Params p) {
    const int b = p.b0 + blockIdx.x;
    ImgState* st = p.st + b;
    if (!st->live) return;
    // Ordered compaction (row-major tile order): neighbouring list entries are neighbouring tiles,
    // whose staged regions overlap, and the priority stage visits neighbouring entries together.
    // A thread takes 4 consecutive tiles per trip (a 4096 x 4096 image has 65536 of them).
    __shared__ unsigned s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nt = p.tnx * p.tny;
    const int* tf = p.tfront + (size_t)b * nt;
    int* tl = p.tlist + ((size_t)b * 2 + st->tlsel) * p.tcap;
    unsigned base = 0;
    for (int t0 = 0; t0 < nt; t0 += 4096) {
        const int t = t0 + 4 * tid;
        int on[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) on[k] = (t + k < nt && tf[imin(t + k, nt - 1)] > 0) ? 1 : 0;
        const unsigned mine = (unsigned)(on[0] + on[1] + on[2] + on[3]);
        // inclusive scan over the wave
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned off = base + incl - mine;
        unsigned total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned c = s_w[w];
            if (w < wv) off += c;
            total += c;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (on[k]) tl[off++] = t + k;
        base += total;
        __syncthreads();
    }
    if (tid == 0) st->tcount = base;
}

// Data term (criminisi_inpainter.py:129-214) and priority (:112) of a front pixel from its new
// confidence, its 3x3 mask neighbourhood and the packed gradient at the patch's first gradient
// maximum.
__device__ __forceinline__ FrontEval finish_front_pixel(double cnew, const int nb[9], int F, int bpk) {
    FrontEval out;
    out.cnew = cnew;
    const double a = nb[0], b = nb[1], c = nb[2], d = nb[3], f = nb[5], g = nb[6], h = nb[7],
                 k = nb[8];
    const double nx = .25 * a - .25 * c + .5 * d - .5 * f + .25 * g - .25 * k;
    const double ny = -.25 * a - .5 * b - .25 * c + .25 * g + .5 * h + .25 * k;
    double nrm = sqrt(ny * ny + nx * nx);
    if (nrm == 0) nrm = 1;
    const double ux = nx / nrm, uy = ny / nrm;
    const int by = (int)(short)(bpk & 0xffff), bx = bpk >> 16;
    const double Gy = (double)by * 0.5, Gx = (double)bx * 0.5;
    const double t0 = ux * Gy, t1 = uy * Gx;
    out.data = sqrt(t0 * t0 + t1 * t1) + 0.001;
    out.prio = out.cnew * out.data * (double)F;
    return out;
}

// Evaluates the n front pixels a priority wave has listed for the tile at (ty0, tx0) from the fields it
// staged in LDS (s_cf: confidence over the tile grown by the patch half-width, RS x RS; s_gk: gradient
// key and packed gradient over the same region, fresh tiles only; s_mask: mask over the tile grown by
// one pixel; s_list: tile-local indices of the front pixels; s_off: the per-lane offset table of the
// fast path).  Groups of PG lanes take one pixel each: new confidence (written to cnxt), data term
// (computed and written to dterm when `fresh`, else read from dterm) and priority; lane 0 of each group
// keeps the best (priority, position) it has seen in bestP / bestI.
template <int PT>
__device__ __forceinline__ void prio_eval_tile(const Params& p, const double* s_cf, const uint2* s_gk,
                                               const uint8_t* s_list, const uint8_t* s_mask, const ushort2* s_off,
                                               const int RS, const int ty0, const int tx0, const int n,
                                               const bool fresh, double* dterm, double* cnxt, const int grp,
                                               const int l, double& bestP, int& bestI, const int ngroups = 64 / PG) {
    constexpr int MS = CRIM_TILE + 2;
    constexpr int NE = PT > 0 ? (PT * PT) / 8 : 1;
    // blocks of 8 in the first half of the pairwise split (all of them when there is no split)
    constexpr int NA = PT * PT <= 128 ? NE : ((PT * PT / 2) - (PT * PT / 2) % 8) / 8;
    static_assert(PT * PT <= 256, "one split level: both halves are leaves");
    const int H = p.H, W = p.W, half = p.half;
    const int ry0 = ty0 - half, rx0 = tx0 - half;
    View<double> vconf; vconf.base = (const unsigned char*)s_cf; vconf.pitch = RS; vconf.oy = ry0; vconf.ox = rx0; vconf.stride = 8;
    View<int> vgrad; vgrad.base = (const unsigned char*)s_gk + 4; vgrad.pitch = RS; vgrad.oy = ry0; vgrad.ox = rx0; vgrad.stride = 8;
    for (int e = grp; e < n; e += ngroups) {   // ngroups: lane groups that share the tile's front pixels
        const int pi = s_list[e];
        const int py = pi >> 4, px = pi & 15;
        const int fy = ty0 + py, fx = tx0 + px;
        const uint8_t* m0 = s_mask + py * MS + px;
        int nb[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) nb[a * 3 + c] = m0[a * MS + c];
        const int F = front_from_nb(nb);
        FrontEval ev;
        const bool inside = fy >= half && fy + half < H && fx >= half && fx + half < W;
        if (!fresh) {
            // cached data term (requested before the sum, used after it)
            const double dcached = dterm[fy * W + fx];
            double cn;
            if (PT > 0 && PG == 4 && inside) {
                const unsigned char* eb = (const unsigned char*)s_cf + (py * RS + px) * 8;
                double r0 = 0.0, r1 = 0.0;
                auto block = [&](int i, bool first) {
                    const ushort2 o = s_off[l * NE + i];
                    const double ca = *(const double*)(eb + o.x);
                    const double cc = *(const double*)(eb + o.y);
                    if (first) { r0 = ca; r1 = cc; } else { r0 += ca; r1 += cc; }
                };
                auto leaf_sum = [&]() {
                    double t = r0 + r1;
                    t = t + dpp_f64<DPP_QUAD_XOR1>(t);
                    return t + dpp_f64<DPP_QUAD_XOR2>(t);
                };
#pragma unroll 5
                for (int i = 0; i < NA; ++i) block(i, i == 0);
                double t = leaf_sum();
                const double tl_ = *(const double*)(eb + ((PT - 1) * RS + (PT - 1)) * 8);
                if constexpr (NA < NE) {
#pragma unroll 5
                    for (int i = NA; i < NE; ++i) block(i, i == NA);
                    double t2 = leaf_sum();
                    t2 += tl_;
                    t = t + t2;
                } else {
                    t += tl_;
                }
                cn = t / (double)p.psq;
            } else {
                cn = front_confidence<(PT > 0 ? (PT * PT > 128 ? 1 : 0) : 3)>(p, vconf, fy, fx, l);
            }
            ev.cnew = cn;
            ev.data = dcached;
            ev.prio = cn * dcached * (double)F;
        } else if (PT > 0 && PG == 4 && inside) {
            const unsigned char* eb = (const unsigned char*)s_cf + (py * RS + px) * 8;
            const unsigned char* gb = (const unsigned char*)s_gk + (py * RS + px) * 8;
            double r0 = 0.0, r1 = 0.0;
            unsigned best = 0;
            const unsigned kb = 1023u - 2u * (unsigned)l;
            auto block = [&](int i, bool first) {
                const ushort2 o = s_off[l * NE + i];
                const double ca = *(const double*)(eb + o.x);
                const double cc = *(const double*)(eb + o.y);
                const unsigned az = *(const unsigned*)(gb + o.x), cz = *(const unsigned*)(gb + o.y);
                if (first) { r0 = ca; r1 = cc; } else { r0 += ca; r1 += cc; }
                const unsigned ka = az + (kb - 8u * i), kc = cz + (kb - 8u * i - 1u);
                best = ka > best ? ka : best;
                best = kc > best ? kc : best;
            };
            // the four lanes' accumulator pairs combined in NumPy's tree order
            auto leaf_sum = [&]() {
                double t = r0 + r1;
                t = t + dpp_f64<DPP_QUAD_XOR1>(t);
                return t + dpp_f64<DPP_QUAD_XOR2>(t);
            };
#pragma unroll 5
            for (int i = 0; i < NA; ++i) block(i, i == 0);
            double t = leaf_sum();
            constexpr int EL = PT * PT - 1;   // the one tail element: PT*PT = 1 (mod 8)
            const double tl_ = *(const double*)(eb + ((PT - 1) * RS + (PT - 1)) * 8);
            const unsigned tlz = *(const unsigned*)(gb + ((PT - 1) * RS + (PT - 1)) * 8);
            if constexpr (NA < NE) {
                // second half of the split: its own leaf, then its tail, then the halves are added
#pragma unroll 5
                for (int i = NA; i < NE; ++i) block(i, i == NA);
                double t2 = leaf_sum();
                t2 += tl_;
                t = t + t2;
            } else {
                t += tl_;
            }
            unsigned ob = (unsigned)dpp_i32<DPP_QUAD_XOR1>((int)best);
            best = ob > best ? ob : best;
            ob = (unsigned)dpp_i32<DPP_QUAD_XOR2>((int)best);
            best = ob > best ? ob : best;
            const unsigned kt = tlz + (1023u - (unsigned)EL);
            best = kt > best ? kt : best;
            const int be = 1023 - (int)(best & 1023u);
            const int brow = be / (PT > 0 ? PT : 1), bcol = be - brow * PT;
            const int bpk = (int)((const uint2*)gb)[brow * RS + bcol].y;
            ev = finish_front_pixel(t / (double)p.psq, nb, F, bpk);
        } else {
            ev = eval_front_pixel<(PT > 0 ? (PT * PT > 128 ? 1 : 0) : 3)>(p, vgrad, vconf, fy, fx, F, nb, l);
        }
        if (l == 0) {
            const int pos = fy * W + fx;
            cnxt[pos] = ev.cnew;
            if (fresh) dterm[pos] = ev.data;
            if (better_prio(ev.prio, pos, bestP, bestI)) { bestP = ev.prio; bestI = pos; }
        }
    }
}

// grid (pgrid, B), one wave per block; block g of image b walks entries g, g + pgrid, ... of
// that image's tile list.  Per tile: the confidence and gradient fields of the tile grown by
// the patch half-width (one 16-byte record per pixel: confidence, 4*|gradient|^2 << 10, packed
// gradient) and the mask grown by one pixel are staged in LDS; front pixels are compacted;
// groups of PG lanes evaluate one front pixel each from LDS.  Every front pixel gets its new
// confidence (written to the other confidence buffer: the update is Jacobi-style, :115-127) and
// its priority; the block emits its best (priority, position).
//
// PT > 0 is the compile-time patch edge of the fast path for pixels whose patch lies inside the
// image: lane l of a group owns accumulators 2l and 2l+1 of NumPy's 8-accumulator leaf, i.e.
// flattened elements 8i + 2l + k, whose LDS offsets relative to the patch corner are the same for
// every pixel and are computed once per kernel.  Up to 128 elements (PT <= 11) the pairwise sum is
// one such leaf; 169 and 225 elements (PT 13, 15) are split once, at a multiple of 8 (80 | 89,
// 112 | 113), and each half is a leaf of its own -- blocks [0, NA) and [NA, NE) of the same element
// numbering -- whose sums are added last.  The same
// elements feed the gradient scan: the largest key (4*|g|^2 << 10) | (1023 - e) is the first
// maximum in flattened order.  Other pixels (clipped patches, larger PT) take the general path.
template <int PT>
__global__ __launch_bounds__(64) void k_priority(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int img_, bx;
    xcd_decode(blockIdx.x, p.pgrid, p.B, &img_, &bx);
    const int b = p.b0 + img_;
    ImgState* st = p.st + b;
    STAMP_MID(0);
    // the first list entry is requested together with the loop state (the list's capacity exceeds
    // the waves per image, so the read is in bounds; it is used only when bx < tcount)
    const int tile_first = p.tlist[((size_t)b * 2 + st->tlsel) * p.tcap + bx];
    const int live = st->live;
    // (k_update rebuilds a list that outgrew its capacity before this kernel runs again; the clamp
    // only keeps a reader inside the list whatever the count says)
    const int nt = imin((int)st->tcount, p.tcap);
    const int cur = st->cur;
    // the patch the previous iteration filled: what changed since this kernel last ran
    const int iter = st->iter;
    const Desc* dlast = p.desc + b;
    const int ur0 = dlast->r0, uc0 = dlast->c0, uph = dlast->ph, upw = dlast->pw;
    if (!live) return;
    const int H = p.H, W = p.W, half = p.half;
    const size_t HW = (size_t)H * W;
    // The data term of a front pixel depends on the mask around it and on the gradient inside its
    // patch, and an iteration changes the mask inside the filled patch and the gradient within one
    // pixel of it: pixels farther than `half` from that box keep their data term, which is cached
    // in p.dterm.  A tile is `fresh` when some pixel of it may need a new data term -- every tile
    // on the first iteration -- and only fresh tiles stage the gradient and run the data-term code.
    const int dy0 = iter == 0 ? 0 : ur0 - 1 - half, dy1 = iter == 0 ? H : ur0 + uph + half;
    const int dx0 = iter == 0 ? 0 : uc0 - 1 - half, dx1 = iter == 0 ? W : uc0 + upw + half;
    double* dterm = p.dterm + b * HW;
    STAMP_MID(1);
    if (bx >= nt) {
        if (threadIdx.x == 0) {
            PEntry e;
            e.prio = -1.0; e.pos = 0x7fffffff; e.pad = 0;
            p.plist[(size_t)b * p.pgrid + bx] = e;
        }
        return;
    }
    const int RS = PT > 0 ? CRIM_TILE + PT - 1 : CRIM_TILE + 2 * half;   // edge of the staged region
    constexpr int MS = CRIM_TILE + 2;                                    // edge of the staged mask
    // Two planes of 8-byte records over the staged region (not one of 16-byte records): the confidence
    // alone is what most tiles read, and consecutive doubles spread over every LDS bank where the
    // confidence halves of 16-byte records use only half of them.
    double* s_cf = (double*)smem;                   // confidence
    uint2* s_gk = (uint2*)(s_cf + RS * RS);         // (4*|gradient|^2 << 10, packed gradient); fresh tiles only
    // (byte and 16-bit entries: at patch edge 9 the block then fits 10 KB, 16 waves per CU)
    uint8_t* s_list = (uint8_t*)(s_gk + RS * RS);   // tile-local pixel index, < 256
    uint8_t* s_mask = s_list + CRIM_TILE * CRIM_TILE;
    const int* tl = p.tlist + ((size_t)b * 2 + st->tlsel) * p.tcap;
    const int* tf = p.tfront + (size_t)b * p.tnx * p.tny;
    const double* ccur = p.conf + ((size_t)b * 2 + cur) * HW;
    double* cnxt = p.conf + ((size_t)b * 2 + (cur ^ 1)) * HW;
    const uint8_t* mask = p.mask + b * HW;
    const int* grad = p.grad + b * HW;
    const int lane = threadIdx.x;
    const int grp = lane / PG, l = lane % PG;
    double bestP = -1.0;
    int bestI = 0x7fffffff;

    // per-lane-of-group table of record offsets (LDS, not registers: the registers buy a fourth
    // wave per SIMD): s_off[l][2i + k] = byte offset of flattened element 8i + 2l + k
    constexpr int NE = PT > 0 ? (PT * PT) / 8 : 1;
    static_assert(CRIM_TILE * CRIM_TILE <= 256, "tile-local pixel indices are bytes");
    // (an offset from the LDS base, not a pointer rounded up through an integer: that would make the
    // table's accesses generic-address loads, which wait for every outstanding global access)
    ushort2* s_off = (ushort2*)(smem + ((RS * RS * 16 + CRIM_TILE * CRIM_TILE + MS * MS + 3) & ~3));
    if (PT > 0) {
        for (int t = lane; t < PG * NE; t += 64) {
            const int ll = t / NE, i = t - ll * NE;
            ushort2 o;
            {
                const int e = 8 * i + 2 * ll;
                const int row = e / (PT > 0 ? PT : 1), col = e - row * PT;
                o.x = (row * RS + col) * 8;
            }
            {
                const int e = 8 * i + 2 * ll + 1;
                const int row = e / (PT > 0 ? PT : 1), col = e - row * PT;
                o.y = (row * RS + col) * 8;
            }
            s_off[ll * NE + i] = o;
        }
    }
    // staging geometry: lanes cover one (RS <= 32) or half a (RS <= 64) staged row per pass
    const int sh = RS <= 32 ? 5 : 6;
    const int rpp = 64 >> sh;
    const int scol = lane & ((1 << sh) - 1), srow = lane >> sh;

    // Rows of staging loads in flight per lane: the whole region when its edge is known at
    // compile time, batches of 16 otherwise.
    constexpr int NRB = PT > 0 ? (CRIM_TILE + PT - 1 + 1) / 2 : 16;
    // The list entries are wave-uniform, and hipcc would wait for such a load on the spot to move the
    // value to a scalar register -- a round trip before the tile's rows are even requested.  Indexed
    // through a zero it cannot see through, the next entry travels with the tile's rows in a vector
    // register and is made scalar once those have arrived.
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    int tile_next = tile_first;
    for (int ti = bx; ti < nt; ti += p.pgrid) {
        if (ti == bx) STAMP_MID(4);
        const int tile = tile_next;
        int tile_next_v = 0;
        // Dependent round trips are what this kernel waits on, so independent loads travel
        // together: the tile's front count (a stale list entry is dropped once it arrives), the
        // next list entry, the mask and the confidence / gradient rows.
        const int tfv = tf[tile];
        if (ti + p.pgrid < nt) tile_next_v = tl[ti + p.pgrid + vzero];
        const int tY = tile / p.tnx, tX = tile - tY * p.tnx;
        const int ty0 = tY * CRIM_TILE, tx0 = tX * CRIM_TILE;
        const int ry0 = ty0 - half, rx0 = tx0 - half;
        // mask grown by one pixel, BORDER_REFLECT_101 beyond the image: two rows of MS columns per
        // sweep of the wave (lanes 0..31 and 32..63; columns MS.. repeat the last one and are dropped)
        constexpr int MSW = (MS + 1) / 2;   // sweeps
        uint8_t mv[MSW];
        {
            auto refl = [](int v, int n) {   // reflect101 of v in [-1, ..): -1 -> 1, >= n -> n - 2
                v = imax(v, -v);
                v = v >= n ? n - 2 : v;
                return imax(v, 0);
            };
            const unsigned mx = (unsigned)refl(tx0 - 1 + imin(lane & 31, MS - 1), W);
#pragma unroll
            for (int k = 0; k < MSW; ++k) {
                const int my = refl(ty0 - 1 + 2 * k + (lane >> 5), H);
                mv[k] = mask[__umul24((unsigned)my, (unsigned)W) + mx];
            }
        }
        // Pixels beyond the image are never read (patches are clipped to it): their records are
        // filled from the clamped position.
        const int xx = imin(imax(rx0 + scol, 0), W - 1);
        const bool fresh = ty0 <= dy1 && ty0 + CRIM_TILE - 1 >= dy0 && tx0 <= dx1 && tx0 + CRIM_TILE - 1 >= dx0;
        bool first = true, stale = false;
        // With a compile-time patch edge the staged region's edge RS is even and a lane's rows srow + 2k
        // (one batch) never pass it: the stores below are then under ONE lane predicate (scol < RS), not
        // one per row -- every separately predicated store costs a save / restore of the execution mask
        // and a branch.
        constexpr bool ROWS_FIT = PT > 0;
        static_assert(PT == 0 || (CRIM_TILE + PT - 1) % 2 == 0, "odd patch edges only");
        for (int rb = srow; rb < RS; rb += NRB * rpp) {
            double cv[NRB];
            int gv[NRB];
            unsigned gidx[NRB];
#pragma unroll
            for (int k = 0; k < NRB; ++k) {
                const int row = imin(rb + k * rpp, RS - 1);
                const int yy = imin(imax(ry0 + row, 0), H - 1);
                // wave-uniform base + 32-bit byte offset (a frame stays far below 2^29 pixels)
                // (24-bit multiply: one full-rate instruction; H, W < 2^24)
                const unsigned idx = __umul24((unsigned)yy, (unsigned)W) + (unsigned)xx;
                cv[k] = *(const double*)((const char*)ccur + (size_t)(idx * 8u));
                gidx[k] = idx * 4u;
            }
            if (fresh) {   // (one wave-uniform branch for all rows)
#pragma unroll
                for (int k = 0; k < NRB; ++k) gv[k] = *(const int*)((const char*)grad + (size_t)gidx[k]);
            }
            if (first) {
                first = false;
                if (tfv <= 0) { stale = true; break; }   // block-uniform
                if ((lane & 31) < MS) {
#pragma unroll
                    for (int k = 0; k < MSW; ++k) s_mask[(2 * k + (lane >> 5)) * MS + (lane & 31)] = mv[k];
                }
            }
            if (fresh) {
                if (scol < RS) {
#pragma unroll
                    for (int k = 0; k < NRB; ++k) {
                        const int row = rb + k * rpp;
                        if (ROWS_FIT || row < RS) {
                            const int gy2 = (int)(short)(gv[k] & 0xffff), gx2 = gv[k] >> 16;
                            s_cf[row * RS + scol] = cv[k];
                            s_gk[row * RS + scol] = make_uint2((uint32_t)(gy2 * gy2 + gx2 * gx2) << 10, (uint32_t)gv[k]);
                        }
                    }
                }
            } else if (scol < RS) {
                // the confidence plane only
#pragma unroll
                for (int k = 0; k < NRB; ++k) {
                    const int row = rb + k * rpp;
                    if (ROWS_FIT || row < RS) s_cf[row * RS + scol] = cv[k];
                }
            }
        }
        tile_next = __builtin_amdgcn_readfirstlane(tile_next_v);
        if (stale) continue;
        __syncthreads();
        if (ti == bx) STAMP_MID(5);
        int n = 0;
        for (int pass = 0; pass < CRIM_TILE * CRIM_TILE / 64; ++pass) {
            const int pi = pass * 64 + lane;
            const int py = pi >> 4, px = pi & 15;
            // (the staged mask covers the whole tile grown by one pixel, also beyond the image: read without
            // a branch, the image bounds enter as a flag)
            const uint8_t* m0 = s_mask + py * MS + px;   // top-left of the 3x3 neighbourhood
            const bool isf = (ty0 + py < H) & (tx0 + px < W) & (m0[MS + 1] == 0) &
                             ((m0[0] + m0[2] + m0[2 * MS] + m0[2 * MS + 2]) > 0);
            const unsigned long long bal = __ballot(isf);
            if (isf) s_list[n + __popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)pi;
            n += __popcll(bal);
        }
        __syncthreads();
        if (ti == bx) STAMP_MID(6);
        prio_eval_tile<PT>(p, s_cf, s_gk, s_list, s_mask, s_off, RS, ty0, tx0, n, fresh, dterm, cnxt, grp, l,
                           bestP, bestI);
        if (ti == bx) STAMP_MID(7);
        __syncthreads();   // the staging buffers are reused by the next tile
    }
    STAMP_MID(2);
    // wave arg-max: highest priority, lowest row-major index on ties (np.argmax, :217)
    wave_best_pair(bestP, bestI, [](double pa, int ia, double pb, int ib) { return better_prio(pa, ia, pb, ib); });
    // every wave owns one slot of its image's partial list (an empty entry has priority -1)
    if (lane == 0) {
        PEntry e;
        e.prio = bestP; e.pos = bestI; e.pad = 0;
        p.plist[(size_t)b * p.pgrid + bx] = e;
    }
    STAMP_MID(3);
}

// Full-frame front / data / priority of the iteration that would run next (debug / parity
// instrument): same arithmetic as k_priority, no state is changed.
__global__ __launch_bounds__(256) void k_fields(Params p, int b, uint8_t* front, double* data,
                                                double* prio) {
    // PG lanes per pixel: a block covers 32 pixels of one tile row pair
    const int l = threadIdx.x % PG, px = threadIdx.x / PG;
    const int y = blockIdx.y * 2 + (px >> 4);
    const int x = blockIdx.x * CRIM_TILE + (px & 15);
    if (y >= p.H || x >= p.W) return;
    const size_t HW = (size_t)p.H * p.W;
    const ImgState* st = p.st + b;
    const double* ccur = p.conf + ((size_t)b * 2 + st->cur) * HW;
    int nb[9];
    const int F = front_weight(p.mask + b * HW, p.H, p.W, y, x, nb);
    const size_t i = (size_t)y * p.W + x;
    double dv = 0.001, pv = 0.0;   // off the front the gradient is zero: sqrt(0) + 0.001
    if (F > 0) {
        View<double> vconf; vconf.base = (const unsigned char*)ccur; vconf.pitch = p.W; vconf.oy = 0; vconf.ox = 0; vconf.stride = 8;
        View<int> vgrad; vgrad.base = (const unsigned char*)(p.grad + b * HW); vgrad.pitch = p.W; vgrad.oy = 0; vgrad.ox = 0; vgrad.stride = 4;
        const FrontEval ev = eval_front_pixel(p, vgrad, vconf, y, x, F, nb, l);
        dv = ev.data;
        pv = ev.prio;
    }
    if (l == 0) {
        if (front) front[i] = (uint8_t)F;
        if (data) data[i] = dv;
        if (prio) prio[i] = pv;
    }
}

// ------------------------------------------------------------------------------------------
// iteration kernel 2: pick the target pixel, build the template descriptor
// ------------------------------------------------------------------------------------------

// NT threads per image: 256 in general; 64 (one wave, NEL = 2 patch elements per lane) for patch edges up to
// 11, where a batch of thousands of images then has every block resident at once.
template <int NT, int NEL>
__global__ __launch_bounds__(NT) void k_select(Params p) {
    const int b = p.b0 + blockIdx.x;
    ImgState* st = p.st + b;
    Desc* d = p.desc + b;
    const int tid = threadIdx.x;
    const long long t_start = tid == 0 ? (long long)wall_clock64() : 0;
    // Dependent memory round trips dominate this kernel, so the partial list is read together with
    // the loop state.
    const PEntry* pl = p.plist + (size_t)b * p.pgrid;
    constexpr int NSPEC = 2;
    PEntry spec[NSPEC];
#pragma unroll
    for (int k = 0; k < NSPEC; ++k) spec[k] = pl[imin(tid + NT * k, p.pgrid - 1)];   // used only when in range
    const int live = st->live;
    const unsigned n = (unsigned)p.pgrid;   // one slot per wave of the priority stage
    const int cur = st->cur;
    const int st_iter = st->iter, st_remaining = st->remaining, st_tlsel = st->tlsel;
    const unsigned st_tcount = st->tcount;
    if (!live) {
        if (tid == 0) d->live = 0;
        return;
    }
    STAMP(8);
    const int H = p.H, W = p.W, P = p.P, ND = p.ND;
    const size_t HW = (size_t)H * W;
    __shared__ double s_bp[4];
    __shared__ int s_bi[4];
    __shared__ int s_target;
    __shared__ uint8_t s_k[CRIM_PMAX * CRIM_PMAX];
    __shared__ int8_t s_t[3][CRIM_PMAX * CRIM_PMAX];
    __shared__ int s_soff;
    __shared__ int s_rowmask;
    __shared__ int s_chmask;
    __shared__ int s_nk;

    double bestP = -1.0;
    int bestI = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NSPEC; ++k)
        if ((unsigned)(tid + NT * k) < n && better_prio(spec[k].prio, spec[k].pos, bestP, bestI)) {
            bestP = spec[k].prio; bestI = spec[k].pos;
        }
    for (unsigned i = tid + NT * NSPEC; i < n; i += NT) {
        const PEntry e = pl[i];
        if (better_prio(e.prio, e.pos, bestP, bestI)) { bestP = e.prio; bestI = e.pos; }
    }
    wave_best_pair(bestP, bestI, [](double pa, int ia, double pb, int ib) { return better_prio(pa, ia, pb, ib); });
    if ((tid & 63) == 0) { s_bp[tid >> 6] = bestP; s_bi[tid >> 6] = bestI; }
    if (tid == 0) { s_soff = 0; s_rowmask = 0; s_chmask = 0; s_nk = 0; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; ++w)
            if (better_prio(s_bp[w], s_bi[w], bestP, bestI)) { bestP = s_bp[w]; bestI = s_bi[w]; }
        // empty front: the priority map is all zeros and np.argmax returns index 0
        s_target = (bestP > 0.0) ? bestI : 0;
    }
    __syncthreads();
    STAMP(9);
    const int pos = s_target;
    const int tr = pos / W, tc = pos - tr * W;
    const int r0 = imax(0, tr - p.half), r1 = imin(tr + p.half, H - 1);
    const int c0 = imax(0, tc - p.half), c1 = imin(tc + p.half, W - 1);
    const int ph = r1 - r0 + 1, pw = c1 - c0 + 1;
    const uint8_t* mask = p.mask + b * HW;
    const uint8_t* planes = p.planes + (size_t)b * CRIM_NPL * H * p.Wp;

    // The target's updated confidence is fetched alongside the patch.  All loads are
    // unconditional, from clamped positions, so that they travel together.
    const double cnew_t = p.conf[((size_t)b * 2 + (cur ^ 1)) * HW + pos];
    const bool mfma_path = P <= MS_PMAX && !p.dot4 && !p.generic;
    // (the launch site picks NT and NEL such that NT * NEL covers the patch)
    int em[NEL], e0[NEL], e1[NEL], e2[NEL];
#pragma unroll
    for (int q = 0; q < NEL; ++q) {
        const int e = imin(tid + NT * q, P * P - 1);
        const int i = e / P, j = e - i * P;
        const int yy = imin(r0 + i, H - 1), xx = imin(c0 + j, W - 1);
        em[q] = mask[(size_t)yy * W + xx];
        const size_t o = (size_t)yy * p.Wp + xx;
        e0[q] = (int8_t)planes[o];
        e1[q] = (int8_t)planes[(size_t)H * p.Wp + o];
        e2[q] = (int8_t)planes[(size_t)2 * H * p.Wp + o];
    }
    // per-thread contributions, reduced across the wave before they touch LDS (same-address LDS
    // atomics from every lane serialise)
    int soff_local = 0, nk_local = 0, rowm_local = 0, chm_local = 0;
#pragma unroll
    for (int q = 0; q < NEL; ++q) {
        const int e = tid + NT * q;
        if (e >= P * P) continue;
        const int i = e / P, j = e - i * P;
        int k = 0, t0 = 0, t1 = 0, t2 = 0;
        if (i < ph && j < pw && !em[q]) {
            k = 1;
            t0 = e0[q]; t1 = e1[q]; t2 = e2[q];
            // sum k*T~^2 - sum k*T^2 with T = T~ + 128
            soff_local += -256 * (t0 + t1 + t2) - 3 * 16384;
            ++nk_local;
            rowm_local |= 1 << i;
            chm_local |= (t0 ? 1 : 0) | (t1 ? 2 : 0) | (t2 ? 4 : 0);
        }
        s_k[e] = (uint8_t)k;
        s_t[0][e] = (int8_t)t0; s_t[1][e] = (int8_t)t1; s_t[2][e] = (int8_t)t2;
        d->kflat[e] = (uint8_t)k;   // also tells the update which pixels of the patch it fills
        if (!mfma_path) {   // flat templates: read by the unspecialised search only
            d->tflat[0][e] = (int8_t)t0; d->tflat[1][e] = (int8_t)t1; d->tflat[2][e] = (int8_t)t2;
        }
    }
    if (__any(nk_local != 0)) {
        soff_local = wave_fold_i32<false>(soff_local);
        nk_local = wave_fold_i32<false>(nk_local);
        rowm_local = wave_fold_i32<true>(rowm_local);
        chm_local = wave_fold_i32<true>(chm_local);
        if ((tid & 63) == 0) {
            atomicAdd(&s_soff, soff_local);
            atomicAdd(&s_nk, nk_local);
            atomicOr(&s_rowmask, rowm_local);
            if (chm_local) atomicOr(&s_chmask, chm_local);
        }
    }
    __syncthreads();
    STAMP(10);

    if (p.dot4) {
        // shifted template rows of the vector search: dword j of shift s covers template columns
        // 4j-s .. 4j-s+3
        const int total = 5 * P * 4 * ND;
        for (int e = tid; e < total; e += NT) {
            const int j = e % ND;
            const int s = (e / ND) & 3;
            const int i = (e / (ND * 4)) % P;
            const int type = e / (ND * 4 * P);
            uint32_t v = 0;
            for (int bb = 0; bb < 4; ++bb) {
                const int col = 4 * j + bb - s;
                uint32_t byte = 0;
                if (col >= 0 && col < P) {
                    const int q = i * P + col;
                    if (type < 3) byte = (uint8_t)s_t[type][q];
                    else if (type == 3) byte = s_k[q];
                    else byte = s_k[q] ? 255u : 0u;
                }
                v |= byte << (8 * bb);
            }
            d->tmpl[e] = v;
        }
    }
    // template rows of the matrix-core search, zero padded to 16 bytes: one row per thread
    if (mfma_path) {
        const int rows = P + 1;
        if (tid < 4 * rows) {
            const int row = tid % rows, type = tid / rows;
            unsigned long long lo = 0, hi = 0;
            if (row < P) {
                const uint8_t* src = type < 3 ? (const uint8_t*)&s_t[type][row * P] : &s_k[row * P];
                for (int c = 0; c < P; ++c) {
                    const unsigned long long byte = src[c];
                    if (c < 8) lo |= byte << (8 * c);
                    else hi |= byte << (8 * (c - 8));
                }
            }
            uint4 o;
            o.x = (uint32_t)lo; o.y = (uint32_t)(lo >> 32); o.z = (uint32_t)hi; o.w = (uint32_t)(hi >> 32);
            *(uint4*)d->zrows[type][row] = o;
        }
    }
    if (tid == 0) {
        const int rowmask = s_rowmask;
        d->live = 1;
        d->tr = tr; d->tc = tc; d->r0 = r0; d->c0 = c0; d->ph = ph; d->pw = pw;
        d->clipped = (ph != P || pw != P) ? 1 : 0;
        d->ncy = H - ph + 1; d->ncx = W - pw + 1;
        d->soff = s_soff;
        d->soff2 = s_soff + 24576 * s_nk;
        d->rowmask = rowmask;
        d->chmask = s_chmask;
        d->cnew = cnew_t;
        d->ty = (double)(r0 + r1) / 2.0;
        d->tx = (double)(c0 + c1) / 2.0;
        d->nfill = ph * pw - s_nk;
        d->cur = cur; d->iter = st_iter; d->remaining = st_remaining;
        d->tlsel = st_tlsel; d->tcount = st_tcount;
        d->t_select = (long long)wall_clock64();
        d->t_entry = t_start;
        st->pcount = 0;
        st->scount = 0;
        st->bound = 0x7fffffff;
    }
    STAMP(11);
}

// ------------------------------------------------------------------------------------------
// iteration kernel 3: masked patch-SSD search
// ------------------------------------------------------------------------------------------

// Does the ph x pw window at origin (y, x) avoid every masked pixel (:250-255)?  The cached
// map answers for full P x P windows; smaller (border-clipped) windows are contained in the
// full window when that fits, and are scanned otherwise.
__device__ __forceinline__ bool window_valid_geom(const Params& p, int clipped, int ph, int pw, const uint8_t* inv,
                                                  const uint8_t* mask, int y, int x) {
    if (y <= p.H - p.P && x <= p.W - p.P && inv[(size_t)y * p.Wi + x] == 0) return true;
    if (!clipped) return false;
    for (int i = 0; i < ph; ++i)
        for (int j = 0; j < pw; ++j)
            if (mask[(size_t)(y + i) * p.W + x + j]) return false;
    return true;
}
__device__ __forceinline__ bool window_valid(const Params& p, const Desc* d, const uint8_t* inv,
                                             const uint8_t* mask, int y, int x) {
    if (y <= p.H - p.P && x <= p.W - p.P && inv[(size_t)y * p.Wi + x] == 0) return true;
    if (!d->clipped) return false;
    for (int i = 0; i < d->ph; ++i)
        for (int j = 0; j < d->pw; ++j)
            if (mask[(size_t)(y + i) * p.W + x + j]) return false;
    return true;
}

__device__ __forceinline__ bool better_cost(double ca, unsigned ia, double cb, unsigned ib) {
    return ca < cb || (ca == cb && ia < ib);
}

// cost = float64(S) + sqrt((y + ph/2 - ty)^2 + (x + pw/2 - tx)^2)   (:258-262)
__device__ __forceinline__ double cand_cost(const Desc* d, int S, int y, int x) {
    const double dy = ((double)y + (double)d->ph / 2.0) - d->ty;
    const double dx = ((double)x + (double)d->pw / 2.0) - d->tx;
    return (double)S + sqrt(dy * dy + dx * dx);
}

// Every search block owns one slot of its image's partial list: block row * snx + block column of
// the launch's tiling (rows are 32 origins for every kernel; the matrix-core kernel's 256-wide tiles
// use the first columns).  A block without an admissible window stores an empty entry, so that
// the consumers can read the slots of the blocks inside the candidate grid without a count.
__device__ __forceinline__ void emit_empty(const Params& p, int b, int slot) {
    SEntry e;
    e.cost = INFINITY; e.idx = 0xffffffffu; e.pad = 0;
    p.slist[(size_t)b * p.snx * p.srows + slot] = e;
}

__device__ __forceinline__ void block_min_emit(const Params& p, int b, int slot, double bc, unsigned bi,
                                               double* s_c, unsigned* s_i) {
    const int tid = threadIdx.x;
    {
        int bi_ = (int)bi;
        wave_best_pair(bc, bi_, [](double ca, int ia, double cb, int ib) { return better_cost(ca, (unsigned)ia, cb, (unsigned)ib); });
        bi = (unsigned)bi_;
    }
    if ((tid & 63) == 0) { s_c[tid >> 6] = bc; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better_cost(s_c[w], s_i[w], bc, bi)) { bc = s_c[w]; bi = s_i[w]; }
        SEntry e;
        e.cost = bc; e.idx = bi; e.pad = 0;
        p.slist[(size_t)b * p.snx * p.srows + slot] = e;
    }
}

// Is entry i of an image's partial list the slot of a block inside the candidate grid (block
// rows [row0, row1), `ncols` block columns of the active tiling)?
__device__ __forceinline__ bool slot_in_grid(const Params& p, int i, int row0, int row1, int ncols) {
    const int r = i / p.snx, c = i - r * p.snx;
    return r >= row0 && r < row1 && c < ncols;
}
__device__ __forceinline__ int search_cols(const Params& p, int ncx) {
    const int tw = (p.generic || p.dot4) ? SB_W : MS_W;
    return (ncx + tw - 1) / tw;
}

// Specialised search: each thread scores 4 consecutive x origins (one dword of every byte
// plane serves all four through pre-shifted templates) for SB_R consecutive rows; plane tiles
// are staged through LDS once per block; template dwords are wave-uniform scalar operands of
// v_dot4.  S_centred = dot(Q%255, k) + dot(Q/255, 255k) - 2 * sum_ch dot(Lab_ch-128, k*T_ch).
template <int P>
__global__ __launch_bounds__(256) void k_search(Params p) {
    constexpr int ND = (P + 6) / 4;
    constexpr int R = SB_R;
    constexpr int LROWS = SB_H + P - 1;
    constexpr int LC4 = (SB_TXG + ND - 1 + 3) / 4;   // 16-byte chunks per staged row
    constexpr int LCOLS = LC4 * 4;
    constexpr int NCHUNK = 5 * LROWS * LC4;
    __shared__ __attribute__((aligned(16))) uint32_t lds[5][LROWS][LCOLS];
    __shared__ double s_c[4];
    __shared__ unsigned s_i[4];
    __shared__ int s_bound;

    const int b = p.b0 + blockIdx.z;
    const Desc* __restrict__ d = p.desc + b;
    if (!d->live) return;
    const int ncy = d->ncy, ncx = d->ncx;
    const int y0 = (p.sy0 + blockIdx.y) * SB_H, x0 = blockIdx.x * SB_W;
    if (y0 >= ncy || x0 >= ncx) return;
    const int tid = threadIdx.x, tx = tid & (SB_TXG - 1), ty = tid / SB_TXG;
    if (tid == 0) s_bound = __hip_atomic_load(&p.st[b].bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int H = p.H, W = p.W;
    const size_t HW = (size_t)H * W;
    const uint8_t* inv = p.invalid + (size_t)b * H * p.Wi;
    const uint8_t* mask = p.mask + b * HW;

    // which of this thread's 4 x R origins are admissible
    unsigned vbits = 0;
    const int xg = x0 + 4 * tx;
    if (!d->clipped) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 + ty * R + r;
            if (y < ncy && xg < ncx) {
                const uint32_t iv = *(const uint32_t*)(inv + (size_t)y * p.Wi + xg);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (xg + s < ncx && ((iv >> (8 * s)) & 0xffu) == 0) vbits |= 1u << (r * 4 + s);
            }
        }
    } else {
        for (int r = 0; r < R; ++r) {
            const int y = y0 + ty * R + r;
            for (int s = 0; s < 4; ++s)
                if (y < ncy && xg + s < ncx && window_valid(p, d, inv, mask, y, xg + s))
                    vbits |= 1u << (r * 4 + s);
        }
    }
    // (Compacting the live 4x4 origin groups so that they fill whole waves was measured: no gain,
    // 57 us vs 56 us per launch of 128 images with a 10 % mask.)
    const int slot = (p.sy0 + (int)blockIdx.y) * p.snx + (int)blockIdx.x;
    if (!__syncthreads_or(vbits != 0)) {
        if (tid == 0) emit_empty(p, b, slot);
        return;
    }

    // A Lab channel whose template k*(T-128) is zero everywhere adds nothing to the cross term
    // (a and b of every grey pixel are exactly 128), so its plane is neither staged nor scored.
    const int chmask = d->chmask | 0x18;
    // stage the byte planes of this tile (zero beyond the image): 16-byte chunks, all of a
    // thread's loads issued before the first LDS store
    const uint8_t* pl = p.planes + (size_t)b * CRIM_NPL * H * p.Wp;
    {
        constexpr int NIT = (NCHUNK + 255) / 256;
        uint4 v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int c = tid + k * 256;
            const int pln = c / (LROWS * LC4);
            const int rem = c - pln * (LROWS * LC4);
            const int row = rem / LC4, c4 = rem - row * LC4;
            const int gy = y0 + row, gx = x0 + 16 * c4;
            v[k] = make_uint4(0, 0, 0, 0);
            if (c < NCHUNK && gy < H && gx + 16 <= p.Wp && ((chmask >> pln) & 1))
                v[k] = *(const uint4*)(pl + ((size_t)pln * H + gy) * p.Wp + gx);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int c = tid + k * 256;
            if (c < NCHUNK) ((uint4*)&lds[0][0][0])[c] = v[k];
        }
    }
    __syncthreads();

    double bc = INFINITY;
    unsigned bi = 0xffffffffu;
    int Sv[R][4];
    int smin = 0x7fffffff, kr = -1, ks = 0;
    if (vbits) {
        int accC[R][4], accQ[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) { accC[r][s] = 0; accQ[r][s] = 0; }
        const_u32_ptr tm = as_const_u32(d->tmpl);
#pragma unroll 1
        for (int ch = 0; ch < 3; ++ch) {
            if (!((chmask >> ch) & 1)) continue;
            uint32_t img[R + P - 1][ND];
#pragma unroll
            for (int rr = 0; rr < R + P - 1; ++rr)
#pragma unroll
                for (int j = 0; j < ND; ++j) img[rr][j] = lds[ch][ty * R + rr][tx + j];
            const_u32_ptr t = tm + ch * (P * 4 * ND);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                // Skipping all-zero template dwords (22 % of them on the 10 %-mask workload) with a
                // wave-uniform branch per 4 dot products was measured 8 % SLOWER: kept branch-free.
                int tv[4][ND];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < ND; ++j) tv[s][j] = (int)t[(i * 4 + s) * ND + j];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < ND; ++j) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            accC[r][s] = __builtin_amdgcn_sdot4((int)img[r + i][j], tv[s][j], accC[r][s], false);
                    }
            }
        }
#pragma unroll 1
        for (int q = 0; q < 2; ++q) {
            uint32_t img[R + P - 1][ND];
#pragma unroll
            for (int rr = 0; rr < R + P - 1; ++rr)
#pragma unroll
                for (int j = 0; j < ND; ++j) img[rr][j] = lds[3 + q][ty * R + rr][tx + j];
            const_u32_ptr t = tm + (3 + q) * (P * 4 * ND);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                uint32_t tv[4][ND];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < ND; ++j) tv[s][j] = t[(i * 4 + s) * ND + j];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < ND; ++j) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            accQ[r][s] = (int)__builtin_amdgcn_udot4(img[r + i][j], tv[s][j], (uint32_t)accQ[r][s], false);
                    }
            }
        }
        const int soff = d->soff;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Sv[r][s] = accQ[r][s] - 2 * accC[r][s] + soff;
                if ((vbits & (1u << (r * 4 + s))) && Sv[r][s] < smin) { smin = Sv[r][s]; kr = r; ks = s; }
            }
    }
    // The distance term is never negative, so a window's cost is at least its integer S.
    // Phase A: every thread prices only its lowest-S window exactly; the smallest integer
    // strictly above any priced cost bounds the minimum from above (block-wide through LDS,
    // grid-wide through a relaxed global word; a stale bound only prunes less).
    // Phase B: the other windows are priced only if their S does not exceed that bound, so no
    // window that could reach or tie the minimum is skipped.
    const int gbound = s_bound;
    if (kr >= 0 && smin <= gbound) {
        const int y = y0 + ty * R + kr, x = xg + ks;
        bc = cand_cost(d, smin, y, x);
        bi = (unsigned)y * (unsigned)ncx + (unsigned)x;
        const double ubd = floor(bc) + 1.0;
        const int ub = ubd < 2147483647.0 ? (int)ubd : 0x7fffffff;
        if (ub < gbound) atomicMin(&s_bound, ub);
    }
    __syncthreads();
    const int ub = s_bound;
    if (vbits) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if ((vbits & (1u << (r * 4 + s))) && !(r == kr && s == ks) && Sv[r][s] <= ub) {
                    const int y = y0 + ty * R + r, x = xg + s;
                    const double c = cand_cost(d, Sv[r][s], y, x);
                    const unsigned idx = (unsigned)y * (unsigned)ncx + (unsigned)x;
                    if (better_cost(c, idx, bc, bi)) { bc = c; bi = idx; }
                }
    }
    if (tid == 0 && ub < gbound) atomicMin(&p.st[b].bound, ub);
    block_min_emit(p, b, slot, bc, bi, s_c, s_i);
}

// Matrix-core search.  For one template row t[0..P) and 16 x shifts n, the 32 x 16 matrix
// B[k][n] = t[k - n] turns the correlation of a 32-byte image segment into a matrix product:
// (A * B)[m][n] = sum_k img[16m + k] * t[k - n] is the row correlation at x origin 16m + n when
// row m of A is the 32 bytes starting at byte 16m.  One v_mfma_i32_16x16x64_i8 handles two
// template rows (K = 2 x 32) for 16 chunks x 16 shifts = 256 consecutive x origins of one origin
// row; the image fragment of rows (r, r+1) serves every (origin row y, step s) with y + 2s = r,
// so each fragment is read from LDS once and used up to (P+1)/2 times.
// Exact integer arithmetic: signed bytes, int32 accumulators.
// With the stored digits 24576 - Q = 256*h + l,
//   -S_reference = 256 * sum k*h + 2 * sum_ch sum (Lab_ch-128) * k*(T_ch-128) + sum k*l - soff2
// is built in ONE accumulator with nothing but left shifts between the passes: the h-digit plane
// first, << 7, plus the cross terms, << 1, plus the l-digit plane; the search then looks for the
// LARGEST accumulator.  Inadmissible windows start from -2^20, i.e. end 2^28 too low.
typedef int v4i __attribute__((ext_vector_type(4)));
#ifndef MS_WPE
#define MS_WPE(P, CHROMA) ((P) > 11 ? 2 : (((CHROMA) || (P) > 9) ? 3 : 4))
#endif

// CHROMA = false serves engines whose images are all grey: only the L channel and the two Q
// digit planes are staged (3 LDS slots instead of 5, twice the blocks per CU).
template <int P, bool CHROMA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MS_WPE(P, CHROMA), MS_WPE(P, CHROMA))))
void k_search_mfma(Params p) {
    constexpr int NS = (P + 1) / 2;
    constexpr int R = MS_R;
    constexpr int NF = R + 2 * NS - 2;          // image fragments per wave and plane
    constexpr int LROWS = MS_ROWS + 2 * NS - 1; // staged rows per plane
    constexpr int C16 = MS_PITCH / 16;          // 16-byte chunks per staged row
    constexpr int NCH = LROWS * C16;
    constexpr int NIT = (NCH + 255) / 256;
    constexpr int POISON = 1 << 20;
    // Three LDS slots hold plane tiles.  Grey engines: L, l digit, h digit, staged once.  Colour
    // engines need five planes: h, L, a are staged first, b and the l digit follow into the
    // slots of h and L while the passes over the resident planes run (three more barriers).
    constexpr int NSLOT = 3;
    constexpr int NCC = CHROMA ? 3 : 1;         // cross-term channels
    constexpr int SLOT_L = 1, SLOT_H = CHROMA ? 0 : 2;
    __shared__ __attribute__((aligned(16))) uint8_t lds[NSLOT][LROWS][MS_PITCH];
    // template rows, [type 0..NCC-1 = k*T per channel, NCC = k][row][48]: 16 zero bytes, the row
    // padded to 16, 16 zero bytes -- a lane's operand bytes are an unaligned 16-byte window of it
    __shared__ __attribute__((aligned(16))) uint8_t zl[NCC + 1][2 * NS][48];
    __shared__ double s_c[4];
    __shared__ unsigned s_i[4];
    __shared__ int s_bound;

    int img_, tile_;
    xcd_decode(blockIdx.x, p.mgx * p.mrows, p.B, &img_, &tile_);
    const int b = p.b0 + img_;
    const Desc* __restrict__ d = p.desc + b;
    if (!d->live) return;
    const int ncy = d->ncy, ncx = d->ncx;
    const int tby = tile_ / p.mgx, tbx = tile_ - tby * p.mgx;
    const int y0 = (p.sy0 + tby) * MS_ROWS, x0 = tbx * MS_W;
    if (y0 >= ncy || x0 >= ncx) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    STAMP_MID(32);
    STAMP_BLOCK(0);
    if (tid == 0) s_bound = __hip_atomic_load(&p.st[b].bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int H = p.H, W = p.W;
    const size_t HW = (size_t)H * W;
    const uint8_t* inv = p.invalid + (size_t)b * H * p.Wi;
    const uint8_t* mask = p.mask + b * HW;
    const int clipped = d->clipped;
    // A channel whose template k*(T-128) is zero everywhere adds nothing to the cross term.
    const int chmask = d->chmask & (CHROMA ? 7 : 1);

    // Accumulator element j of origin row yy in this lane is the window at
    // (y0 + wv*R + yy, x0 + 16*(4*(lane>>4) + j) + (lane&15)).
    const int wx = x0 + 64 * (lane >> 4) + (lane & 15);
    const int wy = y0 + wv * R;

    // ---- every global load of the prologue is issued before the first is used: window validity
    // bytes (addresses clamped into the map, range test applied afterwards), template fragments
    // plane chunks.
    unsigned long long ivw[R];
    if (!clipped) {
        // one 64-bit word per origin row holds this lane's four windows (bits 16j + (lane & 15));
        // origins beyond the candidate grid are marked in the map itself
        const unsigned long long* ib = p.invbits + (size_t)b * H * p.Wb + (x0 >> 6) + (lane >> 4);
#pragma unroll
        for (int yy = 0; yy < R; ++yy) ivw[yy] = ib[(size_t)imin(wy + yy, H - 1) * p.Wb];
    }
    // template rows: one 16-byte chunk per thread, the zero padding by the same threads
    constexpr int NZ = (NCC + 1) * 2 * NS;
    uint4 zv = make_uint4(0, 0, 0, 0);
    if (tid < NZ) {
        const int row = tid % (2 * NS), type = tid / (2 * NS);
        if (row < P && (type == NCC || ((chmask >> type) & 1)))
            zv = *(const uint4*)d->zrows[type == NCC ? 3 : type][row];
    }
    const uint8_t* pl = p.planes + (size_t)b * CRIM_NPL * H * p.Wp;
    int loff[NIT], goff[NIT];
    uint4 v[NSLOT][NIT];
    // plane staged first into each slot
    auto first_plane = [](int s) { return CHROMA ? (s == 0 ? 6 : s - 1) : (s == 0 ? 0 : s + 4); };
    {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int c = tid + k * 256;
            const int row = c / C16, c16 = c - row * C16;
            const int gy = y0 + row, gx = x0 + 16 * c16;
            loff[k] = c < NCH ? row * MS_PITCH + 16 * c16 : -1;
            goff[k] = (c < NCH && gy < H && gx + 16 <= p.Wp) ? gy * p.Wp + gx : -1;
        }
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const uint8_t* src = pl + (size_t)first_plane(s) * H * p.Wp;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                v[s][k] = make_uint4(0, 0, 0, 0);
                if (goff[k] >= 0) v[s][k] = *(const uint4*)(src + goff[k]);
            }
        }
    }

    v4i acc[R];
    int anyvalid = 0;
    if (!clipped) {
        // bit 16j of (word >> n) is window j of the row; an inadmissible window starts at -2^20
        uint32_t all = 0xffffffffu;
#pragma unroll
        for (int yy = 0; yy < R; ++yy) {
            const unsigned long long t = ivw[yy] >> (lane & 15);
            const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
            // one-bit signed field (0 or -1), times 2^20
            acc[yy][0] = __builtin_amdgcn_sbfe((int)lo, 0, 1) << 20;
            acc[yy][1] = __builtin_amdgcn_sbfe((int)lo, 16, 1) << 20;
            acc[yy][2] = __builtin_amdgcn_sbfe((int)hi, 0, 1) << 20;
            acc[yy][3] = __builtin_amdgcn_sbfe((int)hi, 16, 1) << 20;
            all &= lo & hi;
        }
        anyvalid = (all & 0x10001u) != 0x10001u;
    } else {
        for (int yy = 0; yy < R; ++yy)
            for (int j = 0; j < 4; ++j) {
                const int y = wy + yy, x = wx + 16 * j;
                const int ok = (y < ncy && x < ncx && window_valid(p, d, inv, mask, y, x)) ? 1 : 0;
                acc[yy][j] = ok ? 0 : -POISON;
                anyvalid |= ok;
            }
    }
    const int slot = (p.sy0 + tby) * p.snx + tbx;
    if (!__syncthreads_or(anyvalid)) {
        if (tid == 0) emit_empty(p, b, slot);
        return;
    }
    STAMP_MID(33);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (loff[k] >= 0) *(uint4*)(&lds[s][0][0] + loff[k]) = v[s][k];
    }
    if (tid < NZ) {
        uint4* z = (uint4*)&zl[0][0][0] + 3 * tid;
        z[0] = make_uint4(0, 0, 0, 0); z[1] = zv; z[2] = make_uint4(0, 0, 0, 0);
    }
    STAMP_MID(34);
    __syncthreads();
    STAMP_MID(35);

    {
        const bool wave_on = __any(anyvalid);   // a wave without an admissible window only helps staging
        const int g = lane >> 4;
        const int aoff = (wv * R + (g >> 1)) * MS_PITCH + 16 * ((lane & 15) + (g & 1));
        // Operand fragment of step s: lane (n = lane & 15, g) holds, for byte j = 0..15, template
        // row 2s + (g >> 1), column 16 * (g & 1) + j - n (zero outside the patch) -- column n of
        // the Toeplitz matrix -- i.e. the 16 bytes at offset 16 + 16 * (g & 1) - n of the padded row.
        const int zoff = (g >> 1) * 48 + 16 + 16 * (g & 1) - (lane & 15);
        auto fragments = [&](int type, v4i* Bf) {
            const uint8_t* zb = &zl[type][0][0] + (zoff & ~3);
            const int sh = zoff & 3;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t* q = (const uint32_t*)(zb + s * 96);
                const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
                Bf[s][0] = (int)__builtin_amdgcn_alignbyte(w1, w0, sh);
                Bf[s][1] = (int)__builtin_amdgcn_alignbyte(w2, w1, sh);
                Bf[s][2] = (int)__builtin_amdgcn_alignbyte(w3, w2, sh);
                Bf[s][3] = (int)__builtin_amdgcn_alignbyte(w4, w3, sh);
            }
        };
        auto plane_pass = [&](int slot, const v4i* Bf) {
            const uint8_t* base = &lds[slot][0][0] + aoff;
#pragma unroll
            for (int r = 0; r < NF; ++r) {
                const v4i F = *(const v4i*)(base + r * MS_PITCH);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int yy = r - 2 * s;
                    if (yy >= 0 && yy < R)
                        acc[yy] = __builtin_amdgcn_mfma_i32_16x16x64_i8(F, Bf[s], acc[yy], 0, 0, 0);
                }
            }
        };
        v4i Bk[NS], Bc[NS];
        fragments(NCC, Bk);
        if constexpr (!CHROMA) {
            if (wave_on) {
                plane_pass(SLOT_H, Bk);
#pragma unroll
                for (int yy = 0; yy < R; ++yy) acc[yy] <<= 7;
                if (chmask & 1) {
                    fragments(0, Bc);
                    plane_pass(0, Bc);
                }
#pragma unroll
                for (int yy = 0; yy < R; ++yy) acc[yy] <<= 1;
                plane_pass(SLOT_L, Bk);
            }
        } else {
            uint4 vn[NIT];
            auto request = [&](int plane) {
                const uint8_t* src = pl + (size_t)plane * H * p.Wp;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    vn[k] = make_uint4(0, 0, 0, 0);
                    if (goff[k] >= 0) vn[k] = *(const uint4*)(src + goff[k]);
                }
            };
            auto deposit = [&](int slot) {
#pragma unroll
                for (int k = 0; k < NIT; ++k)
                    if (loff[k] >= 0) *(uint4*)(&lds[slot][0][0] + loff[k]) = vn[k];
            };
            if (wave_on) plane_pass(0, Bk);                 // h digits
#pragma unroll
            for (int yy = 0; yy < R; ++yy) acc[yy] <<= 7;
            request(2);                                     // b travels during the L pass
            if (wave_on && (chmask & 1)) {
                fragments(0, Bc);
                plane_pass(1, Bc);                          // L
            }
            __syncthreads();                                // every wave is past the h and L passes
            deposit(0);                                     // b replaces h
            request(5);                                     // the l digits travel during the a pass
            if (wave_on && (chmask & 2)) {
                fragments(1, Bc);
                plane_pass(2, Bc);                          // a
            }
            __syncthreads();                                // b is visible
            deposit(1);                                     // l digits replace L
            if (wave_on && (chmask & 4)) {
                fragments(2, Bc);
                plane_pass(0, Bc);                          // b
            }
            __syncthreads();                                // the l digits are visible
#pragma unroll
            for (int yy = 0; yy < R; ++yy) acc[yy] <<= 1;
            if (wave_on) plane_pass(1, Bk);                 // l digits
        }
    }

    STAMP_MID(36);
    // Pricing.  A window's cost is its integer S plus a distance that is never negative and never
    // more than dmax, so (lowest S of the block) + dmax + 1 bounds the minimum cost from above
    // without pricing anything; the block shares it through LDS and with the other blocks of the
    // image through a relaxed global word (a stale bound only prunes less).  Only windows whose S
    // does not exceed the bound are priced (fp64 sqrt) -- usually one or two per block -- so no
    // window that could reach or tie the minimum is skipped.
    double bc = INFINITY;
    unsigned bi = 0xffffffffu;
    const int soff = d->soff2;            // S_reference = soff - accumulator
    int tmax = -0x7fffffff;
    if (anyvalid) {
#pragma unroll
        for (int yy = 0; yy < R; ++yy)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmax = imax(tmax, acc[yy][j]);
    }
    const int wmax = wave_max_i32(tmax);
    const int gbound = s_bound;
    if (lane == 0 && wmax > -(POISON << 7)) {
        const long long u = (long long)soff - wmax + p.dmax + 1;
        if (u < (long long)gbound) atomicMin(&s_bound, (int)u);
    }
    STAMP_MID(37);
    __syncthreads();
    STAMP_MID(38);
    const int ub = s_bound;
    // windows with S <= ub, i.e. accumulator >= soff - ub (never a poisoned one)
    const long long lim = (long long)soff - ub;
    const int thr = lim < -(long long)(POISON << 7) + 1 ? -(POISON << 7) + 1 : (int)lim;
    if (tmax >= thr) {
#pragma unroll
        for (int yy = 0; yy < R; ++yy)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (acc[yy][j] >= thr) {
                    const int y = wy + yy, x = wx + 16 * j;
                    const double c = cand_cost(d, soff - acc[yy][j], y, x);
                    const unsigned idx = (unsigned)y * (unsigned)ncx + (unsigned)x;
                    if (better_cost(c, idx, bc, bi)) { bc = c; bi = idx; }
                }
    }
    STAMP_MID(39);
    if (tid == 0 && ub < gbound) atomicMin(&p.st[b].bound, ub);
    // the few lanes that priced a window are read out one by one (wave-uniform loop)
    {
        unsigned long long m = __ballot(bi != 0xffffffffu);
        double wc = INFINITY;
        unsigned wi = 0xffffffffu;
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const double c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bc), l),
                                              __builtin_amdgcn_readlane(__double2loint(bc), l));
            const unsigned i = (unsigned)__builtin_amdgcn_readlane((int)bi, l);
            if (better_cost(c, i, wc, wi)) { wc = c; wi = i; }
        }
        bc = wc; bi = wi;
    }
    if (lane == 0) { s_c[wv] = bc; s_i[wv] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better_cost(s_c[w], s_i[w], bc, bi)) { bc = s_c[w]; bi = s_i[w]; }
        SEntry e;
        e.cost = bc; e.idx = bi; e.pad = 0;
        p.slist[(size_t)b * p.snx * p.srows + slot] = e;
    }
    STAMP_MID(40);
}

// Pipelined matrix-core search.  A block walks down up to p.mtiles consecutive row tiles of one
// image column; the planes of a tile (h digits, L, [a, b,] l digits, consumed in that order: 3
// for engines whose images are all grey, 5 otherwise) travel from global memory straight into a
// ring of three LDS slots (global_load_lds, no registers in between) while the passes over the
// other planes run: the slot a pass has just finished with receives the plane needed three
// passes later.  The tile's window-validity words travel the same way with its first plane.
// The template fragments of the k passes stay in registers for all tiles of the block.
//
// Load i of the stream (i = planes-per-tile * tile + pass) goes to slot i % 3 and is issued
// right after the barrier that follows pass i - 3; the barrier before pass i publishes it.
//
// Ordering: a plane is published by `s_waitcnt vmcnt(N)` + s_barrier, where N counts only the
// ring loads certainly issued after it (other outstanding requests make the wait stricter,
// never looser: vector memory operations retire in order).  The barriers are bare s_barrier
// instructions (a __syncthreads() would wait for every outstanding load, the ring's included),
// and the ring loads are issued from inline assembly: the compiler's own bookkeeping of
// LDS-bound loads makes every later LDS read wait for all of them.
__device__ __forceinline__ uint32_t lds_address(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)q;
}
// 64 lanes x 16 (4) bytes from per-lane global addresses to LDS bytes [lds, lds + 1024 (256));
// `lds` is wave-uniform.  M0 is restored: the compiler does not expect it to change.
// The global address is a wave-uniform base plus a per-lane unsigned 32-bit byte offset.
__device__ __forceinline__ void gload_lds16(const void* base, uint32_t off, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void gload_lds4(const void* base, uint32_t off, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(off), "s"(base) : "memory");
}
// Bit `pos` of w as 0 or -1 in one instruction (hipcc expands the one-bit signed field extract into
// an and, a subtraction and shifts).
__device__ __forceinline__ int sbit_field(uint32_t w, int pos) {
    int r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "n"(pos));
    return r;
}
// lgkmcnt(0) and vmcnt(N), N <= 63, in the gfx9 encoding (expcnt left at its maximum)
template <int N>
__device__ __forceinline__ void publish_barrier() {
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | ((N >> 4) << 14));
    __builtin_amdgcn_s_barrier();
}

#ifndef MSP_E2
#define MSP_E2 1
#endif
#ifndef MSP_KEEP_BC
#define MSP_KEEP_BC 0   // (measured: the 12 registers spill at four waves per SIMD, 222 against 187 us per launch)
#endif
#ifndef MSP_E2_DEPTH
#define MSP_E2_DEPTH 3
#endif
#ifndef MSP_SEL_WPE
#define MSP_SEL_WPE(P, CHROMA) 3
#endif
#ifndef MSP_WPE
#define MSP_WPE(P, CHROMA) (((P) > 9 || ((CHROMA) && (P) > 7)) ? 3 : 4)
#endif

// SELECT: the launch also does the work of k_select.  Every block picks the target itself from the
// priority stage's per-wave slots and builds the template rows straight into its LDS (the first
// block of the image also writes the descriptor for k_update): one launch and one round trip less
// per iteration, which is what small batches -- above all a single large image -- wait for.  The
// tile loads are requested between the two steps, so they travel with the target patch's pixels.
// (SELECT instances serve small batches, which never fill four blocks per CU: three waves per SIMD
// give them the registers the extra prologue needs)
template <int P, bool CHROMA, bool SELECT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SELECT ? MSP_SEL_WPE(P, CHROMA) : MSP_WPE(P, CHROMA), SELECT ? MSP_SEL_WPE(P, CHROMA) : MSP_WPE(P, CHROMA))))
void k_search_pipe(Params p) {
    constexpr int NCC = CHROMA ? 3 : 1;         // cross-term channels
    constexpr int NP = NCC + 2;                 // planes (passes) per tile
    constexpr int NS = (P + 1) / 2;
    constexpr int R = MS_R;
    constexpr int NF = R + 2 * NS - 2;
    // E2: the 8 x 2 embedding (patch edges up to 9).  A matrix instruction's 16 columns are 8 x-shifts x
    // 2 y-shifts of a window origin, its 16 rows are image positions 8 bytes apart, and its K = 64 slots are
    // a 4-row x 16-column block of the image: the 16 columns hold 8 shifts + 9 taps - 1 exactly, and NI = 3
    // such blocks hold 2 shifts + 9 taps - 1 = 10 of their 12 rows.  One instruction triple scores 256 windows
    // (128 x-origins x 2 rows) with 192 multiply-adds per window and plane; the Toeplitz embedding of the
    // larger patches (one row pair per instruction, 16 x-shifts + taps in 32 columns) needs 64 NS = 320.
    constexpr bool E2 = MSP_E2 && P <= 9;
    constexpr int NI = (P + 4) / 4;             // E2: row blocks per window pair ((P + 1) / 4, rounded up)
    constexpr int NFR = 2 * NI + 2;             // E2: image row blocks (2 rows apart) a wave's 8 origin rows touch
    constexpr int ZR = E2 ? 4 * NI + 1 : 2 * NS;   // template rows kept in LDS (E2: rows -1 .. 4 NI - 1)
    constexpr int LROWS = E2 ? MS_ROWS + P - 1 : MS_ROWS + 2 * NS - 1;
    constexpr int C16 = MS_PITCH / 16;
    constexpr int POISON = 1 << 20;
    constexpr int NCH = LROWS * C16;            // 16-byte chunks of a plane tile
    constexpr int NK = (NCH + 255) / 256;       // loads per thread and plane (3, or 4 for patch edge 15)
    static_assert(NK <= 4, "a plane tile is at most four 16-byte chunks per thread");
    // A slot holds whole wave loads (64 chunks).  In a thread's last load the waves whose 64 chunks
    // all lie beyond the tile send them to `dump` instead, so that every wave issues NK loads per plane.
    constexpr int SLOT_CHUNKS = (NCH + 63) / 64 * 64;
    constexpr int MSP_SLOT_BYTES = SLOT_CHUNKS * 16;
    __shared__ __attribute__((aligned(16))) uint8_t ring[3][MSP_SLOT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t dump[1024];   // lands the all-padding loads
    __shared__ __attribute__((aligned(16))) unsigned long long vbits[MS_ROWS * 4];
    __shared__ __attribute__((aligned(16))) uint8_t zl[NCC + 1][ZR][48];
    __shared__ double s_c[4];
    __shared__ unsigned s_i[4];
    __shared__ int s_bound;

    int img_, sub_;
    xcd_decode(blockIdx.x, p.mgx * p.mchunks, p.B, &img_, &sub_);
    const int b = p.b0 + img_;
    const Desc* __restrict__ d = p.desc + b;
    // SELECT: k*T and k of the target patch, flattened row-major with row length P
    STAMP_MID(32);
    STAMP_BLOCK(0);
    __shared__ uint8_t s_sk[SELECT ? P * P : 1];
    __shared__ int8_t s_st[SELECT ? 3 : 1][SELECT ? P * P : 1];
    __shared__ double s_sp[SELECT ? 4 : 1];
    __shared__ int s_si[SELECT ? 4 : 1];
    __shared__ int s_red[7];   // soff, nk, rowmask, chmask; SELECT: ph, pw of the target patch (read by the rare clipped path)
    // target geometry worked out by this block (SELECT only)
    struct { int ncy, ncx, clipped, r0, c0, ph, pw; double ty, tx; } sel = {};
    int sel_bound = 0x7fffffff;   // SELECT: the image's pruning bound as it stood when the block started
    if constexpr (!SELECT) {
        if (!d->live) return;
    } else {
        // ---- step 1 of k_select: arg-max over the priority stage's slots (highest priority, lowest
        // row-major index; an empty front leaves the target at pixel 0, :216-218)
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int H = p.H, W = p.W;
        const long long t_start = (sub_ == 0 && tid == 0) ? (long long)wall_clock64() : 0;
        const ImgState* st = p.st + b;
        const PEntry* pl_ = p.plist + (size_t)b * p.pgrid;
        constexpr int NSPEC = 4;
        PEntry spec[NSPEC];
#pragma unroll
        for (int k = 0; k < NSPEC; ++k) spec[k] = pl_[imin(tid + 256 * k, p.pgrid - 1)];
        const int live = st->live;
        const int cur = st->cur;
        const int st_iter = st->iter, st_remaining = st->remaining, st_tlsel = st->tlsel;
        const unsigned st_tcount = st->tcount;
        // (read with the rest of the loop state: a load of its own further down would hold the wave that
        // issues it for a round trip; a stale bound only prunes less)
        sel_bound = st->bound;
        Desc* dw = p.desc + b;   // the first block of the image writes what k_update reads
        if (!live) {
            if (sub_ == 0 && tid == 0) dw->live = 0;
            return;
        }
        double bestP = -1.0;
        int bestI = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < NSPEC; ++k)
            if (tid + 256 * k < p.pgrid && better_prio(spec[k].prio, spec[k].pos, bestP, bestI)) {
                bestP = spec[k].prio; bestI = spec[k].pos;
            }
        for (int i = tid + 256 * NSPEC; i < p.pgrid; i += 256) {
            const PEntry e = pl_[i];
            if (better_prio(e.prio, e.pos, bestP, bestI)) { bestP = e.prio; bestI = e.pos; }
        }
        wave_best_pair(bestP, bestI, [](double pa, int ia, double pb, int ib) { return better_prio(pa, ia, pb, ib); });
        if (lane == 0) { s_sp[wv] = bestP; s_si[wv] = bestI; }
        if (tid < 4) s_red[tid] = 0;
        __syncthreads();
        bestP = s_sp[0]; bestI = s_si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (better_prio(s_sp[w], s_si[w], bestP, bestI)) { bestP = s_sp[w]; bestI = s_si[w]; }
        STAMP_MID(33);
        const int pos = bestP > 0.0 ? bestI : 0;
        const int tr = pos / W, tc = pos - tr * W;
        const int r1 = imin(tr + p.half, H - 1), c1 = imin(tc + p.half, W - 1);
        sel.r0 = imax(0, tr - p.half); sel.c0 = imax(0, tc - p.half);
        sel.ph = r1 - sel.r0 + 1; sel.pw = c1 - sel.c0 + 1;
        sel.clipped = (sel.ph != P || sel.pw != P) ? 1 : 0;
        sel.ncy = H - sel.ph + 1; sel.ncx = W - sel.pw + 1;
        sel.ty = (double)(sel.r0 + r1) / 2.0; sel.tx = (double)(sel.c0 + c1) / 2.0;
        if (tid == 0) { s_red[4] = sel.ph; s_red[5] = sel.pw; }
        if (sub_ == 0 && tid == 0) {
            dw->live = 1;
            dw->tr = tr; dw->tc = tc; dw->r0 = sel.r0; dw->c0 = sel.c0; dw->ph = sel.ph; dw->pw = sel.pw;
            dw->clipped = sel.clipped;
            dw->ncy = sel.ncy; dw->ncx = sel.ncx;
            dw->ty = sel.ty; dw->tx = sel.tx;
            dw->cnew = p.conf[((size_t)b * 2 + (cur ^ 1)) * ((size_t)H * W) + pos];
            dw->cur = cur; dw->iter = st_iter; dw->remaining = st_remaining;
            dw->tlsel = st_tlsel; dw->tcount = st_tcount;
            dw->t_entry = t_start;
        }
    }
    const int ncy = SELECT ? sel.ncy : d->ncy, ncx = SELECT ? sel.ncx : d->ncx;
    const int chunk = sub_ / p.mgx, tbx = sub_ - chunk * p.mgx;
    const int tb0 = p.sy0 + chunk * p.mtiles;          // first row of tiles of this block
    const int x0 = tbx * MS_W;
    if (x0 >= ncx) return;
    int T = imin(p.mtiles, p.mrows - chunk * p.mtiles);
    T = imin(T, (ncy - tb0 * MS_ROWS + MS_ROWS - 1) / MS_ROWS);
    if (T <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = p.H, W = p.W, Wp = p.Wp;
    const size_t HW = (size_t)H * W;
    // everything the pricing needs from the descriptor, read before the first barrier
    const int clipped = SELECT ? sel.clipped : d->clipped;
    int chmask = SELECT ? 0 : (d->chmask & (CHROMA ? 7 : 1));       // SELECT: set with the template rows below
    int soff = SELECT ? 0 : d->soff2;
    const double ph_half = (double)(SELECT ? sel.ph : d->ph) / 2.0, pw_half = (double)(SELECT ? sel.pw : d->pw) / 2.0;
    const double ty = SELECT ? sel.ty : d->ty, tx = SELECT ? sel.tx : d->tx;
    int bound0 = 0;
    if (tid == 0) {
        if constexpr (SELECT) bound0 = sel_bound;
        else bound0 = __hip_atomic_load(&p.st[b].bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_bound = bound0;
    }
    const uint8_t* inv = p.invalid + (size_t)b * H * p.Wi;
    const uint8_t* mask = p.mask + b * HW;
    const int wx = x0 + 64 * (lane >> 4) + (lane & 15);
    // Which window accumulator a, element j of this lane holds.  Toeplitz embedding: origin row a of the wave,
    // x = wx + 16 j.  E2: a = 2 Y + xg, origin row 2 Y + bit 3 of the lane, x = x0 + 128 xg + 32 (lane / 16) +
    // 8 j + (lane & 7).
    const int e2y = (lane >> 3) & 1, e2x = x0 + 32 * (lane >> 4) + (lane & 7);
    auto win_row = [&](int a) { return E2 ? 2 * (a >> 1) + e2y : a; };
    auto win_x = [&](int a, int j) { return E2 ? e2x + 128 * (a & 1) + 8 * j : wx + 16 * j; };

    // ---- ring loads.  Chunk c = tid + 256 k of a tile is row c / C16, bytes 16 (c % C16) ..;
    // columns beyond the plane are read from a clamped address: only windows that are
    // inadmissible anyway see those bytes.  Rows below the image are read as they come (the next
    // plane's or image's rows; the buffer ends in spare rows for the last image).  The offsets
    // are the same for all planes of a tile and advance by 32 rows per tile.
    const uint8_t* pl = p.planes + (size_t)b * CRIM_NPL * H * Wp;
    const size_t plane_bytes = (size_t)H * Wp;
    // plane consumed by pass k of a tile: h digits, L, [a, b,] l digits
    auto plane_of = [](int k) { return k == 0 ? 6 : (k == NP - 1 ? 5 : k - 1); };
    uint32_t coff[NK];            // this tile's chunk offsets
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        // (a chunk beyond the tile -- slot padding, or a load that lands in `dump` -- asks again for the
        // chunk of this thread's first load: a cache hit instead of rows nobody reads)
        const int c = tid + k * 256 < NCH ? tid + k * 256 : tid;
        const int row = c / C16;
        const int gx = imin(x0 + 16 * (c - row * C16), Wp - 16);
        coff[k] = (uint32_t)(tb0 * MS_ROWS + row) * Wp + gx;
    }
    const uint32_t tile_step = (uint32_t)MS_ROWS * Wp;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(lds_address(&ring[0][0]) + 1024 * wv);
    const uint32_t vbits_lds = __builtin_amdgcn_readfirstlane(lds_address(&vbits[0]) + 256 * wv);
    // the last load of a thread: inside the slot, or (this wave's 64 chunks all padding) to `dump`
    const bool last_in_slot = __builtin_amdgcn_readfirstlane((NK - 1) * 256 + 64 * wv < SLOT_CHUNKS);
    const uint32_t last_lds = __builtin_amdgcn_readfirstlane(last_in_slot ? ring_lds + 4096 * (NK - 1) : lds_address(&dump[0]));
    auto load_plane = [&](int plane, const uint32_t* off, int slot) {
        const uint8_t* src = pl + plane * plane_bytes;
#pragma unroll
        for (int k = 0; k < NK - 1; ++k) gload_lds16(src, off[k], ring_lds + slot * MSP_SLOT_BYTES + 4096 * k);
        gload_lds16(src, off[NK - 1], last_lds + (last_in_slot ? slot * MSP_SLOT_BYTES : 0));
    };
    // validity words of a tile: 32 rows x 4 words, one dword per thread
    const unsigned long long* ivb = p.invbits + (size_t)b * H * p.Wb;
    const uint32_t bstep = (uint32_t)p.Wb * 8;
    const uint32_t bmax = (uint32_t)(H - 1) * bstep + ((x0 >> 6) * 8 + (tid & 7) * 4);
    uint32_t boff = umin_u32((uint32_t)(tb0 * MS_ROWS + (tid >> 3)) * bstep + ((x0 >> 6) * 8 + (tid & 7) * 4), bmax);
    auto load_bits = [&](uint32_t off) { gload_lds4(ivb, off, vbits_lds); };

    // SELECT: the target patch's pixels (mask, L, a, b) are requested ahead of the tiles, so that they
    // return first (loads return in order) and the template is built while the tiles are still
    // travelling.  The requests are inline assembly, as the ring loads are: hipcc would count only its
    // own loads and wait for nearly all of the ring's before the first use.
    uint32_t sel_em = 0, sel_e0 = 0, sel_e1 = 0, sel_e2 = 0;
    if constexpr (SELECT) {
        const int e = imin(tid, P * P - 1);
        const int i = e / P, j = e - i * P;
        const int yy = imin(sel.r0 + i, H - 1), xx = imin(sel.c0 + j, W - 1);
        const uint32_t mo = (uint32_t)(yy * W + xx), po = (uint32_t)(yy * Wp + xx);
        asm volatile("global_load_ubyte %0, %1, %2" : "=v"(sel_em) : "v"(mo), "s"(mask) : "memory");
        asm volatile("global_load_sbyte %0, %1, %2" : "=v"(sel_e0) : "v"(po), "s"(pl) : "memory");
        asm volatile("global_load_sbyte %0, %1, %2" : "=v"(sel_e1) : "v"(po), "s"(pl + plane_bytes) : "memory");
        asm volatile("global_load_sbyte %0, %1, %2" : "=v"(sel_e2) : "v"(po), "s"(pl + 2 * plane_bytes) : "memory");
    }
    // prologue: the first three planes of the first tile; template rows into LDS
    load_plane(plane_of(0), coff, 0);
    load_bits(boff);
    load_plane(plane_of(1), coff, 1);
    load_plane(plane_of(2), coff, 2);
    if constexpr (SELECT) {
        // the four patch loads have returned once at most the 3 NK + 1 ring loads behind them are outstanding
        constexpr int BEHIND = 3 * NK + 1;
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(sel_em), "+v"(sel_e0), "+v"(sel_e1), "+v"(sel_e2) : "n"(BEHIND) : "memory");
    }
    constexpr int NZ = (NCC + 1) * ZR;
    static_assert(NZ <= 256, "one template row per thread");
    if constexpr (!SELECT) {
        if (tid < NZ) {
            const int row = tid % ZR - (E2 ? 1 : 0), type = tid / ZR;
            uint4 zv = make_uint4(0, 0, 0, 0);
            if (row >= 0 && row < P && (type == NCC || ((chmask >> type) & 1)))
                zv = *(const uint4*)d->zrows[type == NCC ? 3 : type][row];
            uint4* z = (uint4*)&zl[0][0][0] + 3 * tid;
            z[0] = make_uint4(0, 0, 0, 0); z[1] = zv; z[2] = make_uint4(0, 0, 0, 0);
        }
    } else {
        // ---- step 2 of k_select: k = 1 - mask and k * (Lab - 128) over the target patch (:227-235),
        // read while the tile loads above travel
        static_assert(P * P <= 256, "one patch element per thread");
        {
            const int e = imin(tid, P * P - 1);
            const int i = e / P, j = e - i * P;
            const int em = (int)sel_em;
            const int e0 = (int)sel_e0, e1 = (int)sel_e1, e2 = (int)sel_e2;
            int soff_l = 0, nk_l = 0, rowm_l = 0, chm_l = 0;
            if (tid < P * P) {
                int k = 0, t0 = 0, t1 = 0, t2 = 0;
                if (i < sel.ph && j < sel.pw && !em) {
                    k = 1;
                    t0 = e0; t1 = e1; t2 = e2;
                    soff_l = -256 * (t0 + t1 + t2) - 3 * 16384;   // sum k*T~^2 - sum k*T^2 with T = T~ + 128
                    nk_l = 1;
                    rowm_l = 1 << i;
                    chm_l = (t0 ? 1 : 0) | (t1 ? 2 : 0) | (t2 ? 4 : 0);
                }
                s_sk[tid] = (uint8_t)k;
                s_st[0][tid] = (int8_t)t0; s_st[1][tid] = (int8_t)t1; s_st[2][tid] = (int8_t)t2;
                if (sub_ == 0) (p.desc + b)->kflat[tid] = (uint8_t)k;   // tells the update which pixels it fills
            }
            soff_l = wave_fold_i32<false>(soff_l);
            nk_l = wave_fold_i32<false>(nk_l);
            rowm_l = wave_fold_i32<true>(rowm_l);
            chm_l = wave_fold_i32<true>(chm_l);
            if (lane == 0 && nk_l) {
                atomicAdd(&s_red[0], soff_l);
                atomicAdd(&s_red[1], nk_l);
                atomicOr(&s_red[2], rowm_l);
                atomicOr(&s_red[3], chm_l);
            }
        }
        publish_barrier<63>();                 // LDS only: the tile loads stay in flight
        soff = s_red[0] + 24576 * s_red[1];
        chmask = s_red[3] & (CHROMA ? 7 : 1);
        if (sub_ == 0 && tid == 0) {
            (p.desc + b)->nfill = sel.ph * sel.pw - s_red[1];
            (p.desc + b)->t_select = (long long)wall_clock64();
        }
        if (tid < NZ) {
            const int row = tid % ZR - (E2 ? 1 : 0), type = tid / ZR;
            unsigned long long lo = 0, hi = 0;
            if (row >= 0 && row < P && (type == NCC || ((chmask >> type) & 1))) {
                const uint8_t* src = type == NCC ? &s_sk[row * P] : (const uint8_t*)&s_st[type][row * P];
#pragma unroll
                for (int c = 0; c < P; ++c) {
                    const unsigned long long byte = src[c];
                    if (c < 8) lo |= byte << (8 * c);
                    else hi |= byte << (8 * (c - 8));
                }
            }
            uint4 zv;
            zv.x = (uint32_t)lo; zv.y = (uint32_t)(lo >> 32); zv.z = (uint32_t)hi; zv.w = (uint32_t)(hi >> 32);
            uint4* z = (uint4*)&zl[0][0][0] + 3 * tid;
            z[0] = make_uint4(0, 0, 0, 0); z[1] = zv; z[2] = make_uint4(0, 0, 0, 0);
        }
    }

    STAMP_MID(34);
    const int g = lane >> 4;
    // E2: K-group g of a matrix instruction holds rows 2 (g / 2), 2 (g / 2) + 1 of the 4-row block and its
    // columns 8 (g & 1) .. + 7: the lanes of groups 0 and 1 (2 and 3) then read the same rows 8 bytes apart,
    // lane (m, 1) the very quadword of lane (m + 1, 0), so a half wave's 64-bit reads touch every bank once.
    const int aoff = E2 ? (wv * R + 2 * (g >> 1)) * MS_PITCH + 8 * ((lane & 15) + (g & 1))
                        : (wv * R + (g >> 1)) * MS_PITCH + 16 * ((lane & 15) + (g & 1));
    // E2: the last row block of the last wave ends below the staged rows; its lower half only meets template
    // rows beyond the patch (all zero), so those lanes read two rows higher instead
    constexpr bool E2_LAST_CLAMPED = E2 && (MS_ROWS - R + 2 * (NFR - 1) + 3 >= LROWS);
    const int aoff_last = (E2_LAST_CLAMPED && g >= 2) ? aoff - 2 * MS_PITCH : aoff;
    const int zoff = E2 ? (2 * (g >> 1) - e2y + 1) * 48 + 16 + 8 * (g & 1) - (lane & 7)
                        : (g >> 1) * 48 + 16 + 16 * (g & 1) - (lane & 15);
    auto fragments = [&](int type, v4i* Bf) {
        const uint8_t* zb = &zl[type][0][0] + (zoff & ~3);
        const int sh = zoff & 3;
        if constexpr (E2) {
            // row block i, rows 4 i + 2 (g / 2) + {0, 1} - (y-shift) of the template, 8 columns from
            // 8 (g & 1) - (x-shift): two unaligned 8-byte windows of the zero-padded rows
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t* q0 = (const uint32_t*)(zb + (4 * i) * 48);
                const uint32_t* q1 = (const uint32_t*)(zb + (4 * i + 1) * 48);
                const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2];
                const uint32_t b0 = q1[0], b1 = q1[1], b2 = q1[2];
                Bf[i][0] = (int)__builtin_amdgcn_alignbyte(a1, a0, sh);
                Bf[i][1] = (int)__builtin_amdgcn_alignbyte(a2, a1, sh);
                Bf[i][2] = (int)__builtin_amdgcn_alignbyte(b1, b0, sh);
                Bf[i][3] = (int)__builtin_amdgcn_alignbyte(b2, b1, sh);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t* q = (const uint32_t*)(zb + s * 96);
                const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
                Bf[s][0] = (int)__builtin_amdgcn_alignbyte(w1, w0, sh);
                Bf[s][1] = (int)__builtin_amdgcn_alignbyte(w2, w1, sh);
                Bf[s][2] = (int)__builtin_amdgcn_alignbyte(w3, w2, sh);
                Bf[s][3] = (int)__builtin_amdgcn_alignbyte(w4, w3, sh);
            }
        }
    };
    constexpr int NB = E2 ? NI : NS;            // template fragments per plane kind
    v4i Bk[NB];
    // grey engines with the 8 x 2 embedding keep the one cross-term channel's fragments as well
    constexpr bool KEEP_BC = MSP_KEEP_BC && E2 && !CHROMA;
    v4i Bc0[KEEP_BC ? NB : 1];
    v4i acc[R];
    // Toeplitz embedding: two row registers, refilled alternately: the read of row r + 2 is issued right after
    // the matrix instructions of row r, so its LDS latency passes under those of row r + 1 (left to itself
    // hipcc reads rows in pairs and waits for each pair with nothing in flight).
    // E2: image fragment (f, xg) = rows 2 f .. 2 f + 3 of the wave's rows, columns 128 xg ..; it serves the
    // window pairs Y with row block i = (f - Y) / 2.  MSP_E2_DEPTH fragment registers, all but one being read.
    auto plane_pass = [&](int slot, const v4i* Bf) {
        if constexpr (E2) {
            const uint8_t* base = &ring[slot][0];
            auto read = [&](int k) {
                const int f = k >> 1, xg = k & 1;
                const int off = (E2_LAST_CLAMPED && f == NFR - 1) ? aoff_last : aoff;
                const unsigned long long* q = (const unsigned long long*)(base + off + f * 2 * MS_PITCH + 128 * xg);
                const unsigned long long lo = q[0], hi = q[MS_PITCH / 8];
                return v4i{(int)(uint32_t)lo, (int)(uint32_t)(lo >> 32), (int)(uint32_t)hi, (int)(uint32_t)(hi >> 32)};
            };
            auto use = [&](int k, const v4i& F) {
                const int f = k >> 1, xg = k & 1;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int Y = f - 2 * i;
                    if (Y >= 0 && Y < R / 2)
                        acc[2 * Y + xg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(F, Bf[i], acc[2 * Y + xg], 0, 0, 0);
                }
            };
            constexpr int NK2 = 2 * NFR;
            constexpr int D = MSP_E2_DEPTH;     // fragment registers: D - 1 reads in flight
            v4i F[D];
#pragma unroll
            for (int k = 0; k < D - 1; ++k) F[k] = read(k);
#pragma unroll
            for (int k = 0; k < NK2; ++k) {
                __builtin_amdgcn_sched_barrier(0);
                if (k + D - 1 < NK2) F[(k + D - 1) % D] = read(k + D - 1);
                __builtin_amdgcn_sched_barrier(0);
                use(k, F[k % D]);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
        const uint8_t* base = &ring[slot][0] + aoff;
        static_assert(NF % 2 == 0, "rows are consumed in pairs");
        v4i Fa = *(const v4i*)(base), Fb = *(const v4i*)(base + MS_PITCH);
        auto row = [&](int r, const v4i& F) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int yy = r - 2 * s;
                if (yy >= 0 && yy < R)
                    acc[yy] = __builtin_amdgcn_mfma_i32_16x16x64_i8(F, Bf[s], acc[yy], 0, 0, 0);
            }
        };
#pragma unroll
        for (int r = 0; r < NF; r += 2) {
            __builtin_amdgcn_sched_barrier(0);
            row(r, Fa);
            __builtin_amdgcn_sched_barrier(0);
            if (r + 2 < NF) Fa = *(const v4i*)(base + (r + 2) * MS_PITCH);
            __builtin_amdgcn_sched_barrier(0);
            row(r + 1, Fb);
            __builtin_amdgcn_sched_barrier(0);
            if (r + 3 < NF) Fb = *(const v4i*)(base + (r + 3) * MS_PITCH);
        }
        __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the block's lowest cost of a tile is written by thread 0 one barrier after the waves posted theirs
    auto store_result = [&](int tb) {
        double bc = s_c[0];
        unsigned bi = s_i[0];
        for (int w = 1; w < 4; ++w)
            if (better_cost(s_c[w], s_i[w], bc, bi)) { bc = s_c[w]; bi = s_i[w]; }
        SEntry e;
        e.cost = bc; e.idx = bi; e.pad = 0;
        p.slist[(size_t)b * p.snx * p.srows + tb * p.snx + tbx] = e;
    };

    // ring slot of stream load NP * t + i: (tbase + i) % 3 with tbase = (NP * t) % 3
    int tbase = 0;
    auto slot_of = [&](int i) {
        int s = tbase + i % 3;
        return s >= 3 ? s - 3 : s;
    };
    auto publish = [](int n) {   // n is a compile-time constant after unrolling
        switch (n) {
            case 0: publish_barrier<0>(); break;
            case 3: publish_barrier<3>(); break;
            case 4: publish_barrier<4>(); break;
            case 5: publish_barrier<5>(); break;
            case 6: publish_barrier<6>(); break;
            case 8: publish_barrier<8>(); break;
            default: publish_barrier<63>(); break;
        }
    };
    // A wave is `clean` in a tile when every window of its rows left of the image's right edge is
    // admissible and that edge cuts at most the last of a lane's four windows: its accumulators then
    // start from zero (no start values worked out per window) and the edge windows are taken out
    // after the passes.
    const int edge_lim = ncx - x0;                       // columns [edge_lim, 256) of the tile lie beyond the edge
    const bool edge_ok = !clipped && edge_lim >= 240;
    const unsigned long long edge_word = edge_lim >= 256 ? 0ull : ~0ull << (imax(edge_lim, 192) - 192);
    for (int t = 0; t < T; ++t) {
        const unsigned long long* vb = vbits;
        const int tb = tb0 + t;
        const int wy = tb * MS_ROWS + wv * R;
        const bool more = t + 1 < T;
        bool wave_on = false;
        bool clean = false;
        // (diagnostic build: the phases of the block's fourth tile, slots 40..52; tools/search_tile_stamps.py)
        TSTAMP(t == 3, 40);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            // ---- barrier before pass j: load j is waited for, counting the ring loads issued after
            // it (loads 1 and 2 before pass 0; load j + 1 otherwise, one more operation when it carries
            // the next tile's validity words)
            if (j == 0) publish(2 * NK);
            else if (j + 1 < NP) publish(NK);
            else if (more) publish(NK + 1);
            else publish(0);
            TSTAMP(t == 3, 41 + 3 * (j < 3 ? j : 2));
            if (j >= 1) {
                // pass j - 1 is over in every wave: its slot takes load j + 2
                const int i = j + 2;
                if (i < NP) {
                    load_plane(plane_of(i), coff, slot_of(i));
                } else if (more) {
                    if (i == NP) {
#pragma unroll
                        for (int k = 0; k < NK; ++k) coff[k] += tile_step;
                        boff = umin_u32(boff + MS_ROWS * bstep, bmax);
                    }
                    load_plane(plane_of(i - NP), coff, slot_of(i));
                    if (i == NP) load_bits(boff);
                }
            }
            if (j == 0) {
                if (t == 0) STAMP_MID(35);
                if (t > 0 && tid == 0) store_result(tb - 1);
                if (t == 0) {
                    fragments(NCC, Bk);
                    if constexpr (KEEP_BC) fragments(0, Bc0);
                }
                // every wave looks at the whole tile's validity words: is there any admissible window?
                const uint4 q = ((const uint4*)vb)[lane];
                const bool tile_any = clipped || __ballot((q.x & q.y & q.z & q.w) != 0xffffffffu) != 0;
                bool wave_any;
                if (!clipped) {
                    // the wave's 8 rows of words: lanes 0..15 look at 16 bytes each
                    const uint4 qw = ((const uint4*)vb)[wv * 16 + (lane & 15)];
                    wave_any = __ballot((qw.x & qw.y & qw.z & qw.w) != 0xffffffffu) != 0;
                    if (edge_ok) {
                        // even lanes hold words 0 and 1 of a row, odd lanes words 2 and 3
                        // (three ballots against scalar operands: nothing lane-dependent is kept across tiles)
                        const unsigned long long off02 = __ballot((qw.x | qw.y) != 0u);
                        const unsigned long long off1 = __ballot((qw.z | qw.w) != 0u) & 0x5555555555555555ull;
                        const unsigned long long off3 = __ballot(qw.z != (uint32_t)edge_word || qw.w != (uint32_t)(edge_word >> 32)) & 0xaaaaaaaaaaaaaaaaull;
                        clean = (off02 | off1 | off3) == 0;
                    }
                    if (clean) {
                        // (a zero the compiler cannot see through: folded into the first matrix instruction of
                        // each accumulator it would cost the in-place update of the others their registers)
                        int zero;
                        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
                        for (int yy = 0; yy < R; ++yy) acc[yy] = v4i{zero, zero, zero, zero};
                    } else {
#pragma unroll
                        for (int yy = 0; yy < R; ++yy) {
                            if constexpr (E2) {
                                // windows 32 (g & 1) + 8 j + (lane & 7) of word 2 xg + g / 2
                                const unsigned long long w = vb[(wv * R + win_row(yy)) * 4 + 2 * (yy & 1) + (g >> 1)];
                                const uint32_t lo = ((g & 1) ? (uint32_t)(w >> 32) : (uint32_t)w) >> (lane & 7);
                                acc[yy][0] = sbit_field(lo, 0) << 20;
                                acc[yy][1] = sbit_field(lo, 8) << 20;
                                acc[yy][2] = sbit_field(lo, 16) << 20;
                                acc[yy][3] = sbit_field(lo, 24) << 20;
                            } else {
                                const unsigned long long w = vb[(wv * R + yy) * 4 + g] >> (lane & 15);
                                const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
                                acc[yy][0] = sbit_field(lo, 0) << 20;
                                acc[yy][1] = sbit_field(lo, 16) << 20;
                                acc[yy][2] = sbit_field(hi, 0) << 20;
                                acc[yy][3] = sbit_field(hi, 16) << 20;
                            }
                        }
                    }
                } else {
                    int anyvalid = 0;
                    for (int yy = 0; yy < R; ++yy)
                        for (int jj = 0; jj < 4; ++jj) {
                            const int y = wy + win_row(yy), x = win_x(yy, jj);
                            // (only origins inside the candidate grid are looked at: the mask test reads the
                            // window's pixels)
                            bool wok = y < ncy && x < ncx;
                            if (wok) {
                                if constexpr (SELECT) wok = window_valid_geom(p, 1, s_red[1 + 3], s_red[2 + 3], inv, mask, y, x);
                                else wok = window_valid(p, d, inv, mask, y, x);
                            }
                            const int ok = wok ? 1 : 0;
                            acc[yy][jj] = ok ? 0 : -POISON;
                            anyvalid |= ok;
                        }
                    wave_any = __any(anyvalid);
                }
                wave_on = tile_any && wave_any;
            }
            TSTAMP(t == 3, 42 + 3 * (j < 3 ? j : 2));
            // ---- pass j
            if (j == 1) {
#pragma unroll
                for (int yy = 0; yy < R; ++yy) acc[yy] <<= 7;
            }
            if (j == NP - 1) {
#pragma unroll
                for (int yy = 0; yy < R; ++yy) acc[yy] <<= 1;
            }
            if (j == 0 || j == NP - 1) {
                if (wave_on) plane_pass(slot_of(j), Bk);
            } else if (wave_on && ((chmask >> (j - 1)) & 1)) {
                if constexpr (KEEP_BC) {
                    plane_pass(slot_of(j), Bc0);
                } else {
                    // the cross-term fragments are rebuilt for every tile: keeping them would cost 20
                    // registers per channel (12 with the 8 x 2 embedding) and the fourth wave per SIMD
                    v4i Bc[NB];
                    fragments(j - 1, Bc);
                    plane_pass(slot_of(j), Bc);
                }
            }
            TSTAMP(t == 3, 43 + 3 * (j < 3 ? j : 2));
        }

        if (clean) {
            // windows beyond the right edge, as far below as a start value of -POISON would have put them
            if constexpr (E2) {
                // (only the tile's last 16 columns can lie beyond the edge: elements 2 and 3 of the right halves)
                const int add2 = win_x(1, 2) >= ncx ? -(POISON << 8) : 0;
                const int add3 = win_x(1, 3) >= ncx ? -(POISON << 8) : 0;
#pragma unroll
                for (int yy = 1; yy < R; yy += 2) { acc[yy][2] += add2; acc[yy][3] += add3; }
            } else {
                const int edge_add = wx + 48 >= ncx ? -(POISON << 8) : 0;
#pragma unroll
                for (int yy = 0; yy < R; ++yy) acc[yy][3] += edge_add;
            }
        }
        if (t == 0) STAMP_MID(36);
        // pricing, as in k_search_mfma
        double bc = INFINITY;
        unsigned bi = 0xffffffffu;
        // an inadmissible window ends 2^28 too low, far below any admissible one
        int tmax = -0x7fffffff;
#pragma unroll
        for (int yy = 0; yy < R; ++yy)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmax = imax(tmax, acc[yy][j]);
        const int wmax = wave_max_i32(tmax);
        if (lane == 0 && wmax > -(POISON << 7)) {
            const long long u = (long long)soff - wmax + p.dmax + 1;
            if (u < (long long)__hip_atomic_load(&s_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                atomicMin(&s_bound, (int)u);
        }
        TSTAMP(t == 3, 50);
        publish_barrier<63>();                      // bounds posted; no ring load is waited for
        TSTAMP(t == 3, 51);
        if (more) load_plane(plane_of(2), coff, slot_of(NP + 2));   // the last pass's slot takes load NP + 2
        if (t == 0) STAMP_MID(37);
        const int ub = __hip_atomic_load(&s_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const long long lim = (long long)soff - ub;
        const int thr = lim < -(long long)(POISON << 7) + 1 ? -(POISON << 7) + 1 : (int)lim;
        if (tmax >= thr) {
#pragma unroll
            for (int yy = 0; yy < R; ++yy)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (acc[yy][j] >= thr) {
                        const int y = wy + win_row(yy), x = win_x(yy, j);
                        // cand_cost with the descriptor fields read up front
                        const double dy = ((double)y + ph_half) - ty;
                        const double dx = ((double)x + pw_half) - tx;
                        const double c = (double)(soff - acc[yy][j]) + sqrt(dy * dy + dx * dx);
                        const unsigned idx = (unsigned)y * (unsigned)ncx + (unsigned)x;
                        if (better_cost(c, idx, bc, bi)) { bc = c; bi = idx; }
                    }
        }
        {
            unsigned long long m = __ballot(bi != 0xffffffffu);
            double wc = INFINITY;
            unsigned wi = 0xffffffffu;
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const double c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bc), l),
                                                  __builtin_amdgcn_readlane(__double2loint(bc), l));
                const unsigned i = (unsigned)__builtin_amdgcn_readlane((int)bi, l);
                if (better_cost(c, i, wc, wi)) { wc = c; wi = i; }
            }
            if (lane == 0) { s_c[wv] = wc; s_i[wv] = wi; }
        }
        tbase += NP % 3;
        if (tbase >= 3) tbase -= 3;
        TSTAMP(t == 3, 52);
    }
    publish_barrier<0>();
    if (tid == 0) {
        store_result(tb0 + T - 1);
        const int ub = s_bound;
        if (ub < bound0) atomicMin(&p.st[b].bound, ub);
    }
    STAMP_MID(38);
    STAMP_BLOCK(1);
}

// Unspecialised search for any patch edge: one origin at a time, planes read from global
// memory, templates from the flat byte arrays.
__global__ __launch_bounds__(256) void k_search_generic(Params p) {
    __shared__ double s_c[4];
    __shared__ unsigned s_i[4];
    const int b = p.b0 + blockIdx.z;
    const Desc* __restrict__ d = p.desc + b;
    if (!d->live) return;
    const int ncy = d->ncy, ncx = d->ncx;
    const int y0 = (p.sy0 + blockIdx.y) * SB_H, x0 = blockIdx.x * SB_W;
    if (y0 >= ncy || x0 >= ncx) return;
    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, P = p.P;
    const size_t HW = (size_t)H * W;
    const uint8_t* inv = p.invalid + (size_t)b * H * p.Wi;
    const uint8_t* mask = p.mask + b * HW;
    const uint8_t* pl = p.planes + (size_t)b * CRIM_NPL * H * p.Wp;
    const size_t PS = (size_t)H * p.Wp;
    const int ph = d->ph, pw = d->pw, soff = d->soff;
    double bc = INFINITY;
    unsigned bi = 0xffffffffu;
    for (int e = tid; e < SB_W * SB_H; e += 256) {
        const int y = y0 + e / SB_W, x = x0 + e % SB_W;
        if (y >= ncy || x >= ncx) continue;
        if (!window_valid(p, d, inv, mask, y, x)) continue;
        int accQ = 0, accC = 0;
        for (int i = 0; i < ph; ++i)
            for (int j = 0; j < pw; ++j) {
                const int q = i * P + j;
                if (!d->kflat[q]) continue;
                const size_t o = (size_t)(y + i) * p.Wp + x + j;
                accQ += (int)pl[3 * PS + o] + 255 * (int)pl[4 * PS + o];
                accC += (int)(int8_t)pl[o] * (int)d->tflat[0][q] +
                        (int)(int8_t)pl[PS + o] * (int)d->tflat[1][q] +
                        (int)(int8_t)pl[2 * PS + o] * (int)d->tflat[2][q];
            }
        const int S = accQ - 2 * accC + soff;
        if ((double)S <= bc) {
            const double c = cand_cost(d, S, y, x);
            const unsigned idx = (unsigned)y * (unsigned)ncx + (unsigned)x;
            if (better_cost(c, idx, bc, bi)) { bc = c; bi = idx; }
        }
    }
    block_min_emit(p, b, (p.sy0 + (int)blockIdx.y) * p.snx + (int)blockIdx.x, bc, bi, s_c, s_i);
}

// ------------------------------------------------------------------------------------------
// iteration kernel 4: pick the source, copy the patch, maintain every cached field
// ------------------------------------------------------------------------------------------

// PM bounds the patch edge the instance serves: its LDS arrays are sized for it, so the common small
// patches leave room for many blocks per CU.
// Patch edges up to 9: eight waves per SIMD (the scalar registers are capped for it), so that 2048
// blocks, one per image, are resident at once.
template <int PM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PM <= 9 ? 8 : 1, 8))) void k_update(Params p) {
    const int b = p.b0 + blockIdx.x;
    ImgState* st = p.st + b;
    const Desc* __restrict__ d = p.desc + b;
    const int tid = threadIdx.x;
    const long long t_entry = tid == 0 ? (long long)wall_clock64() : 0;   // the search ended about here
    const int H = p.H, W = p.W, P = p.P;
    const size_t HW = (size_t)H * W;
    constexpr int MR = 3 * PM - 2 + 2;  // mask region edge: patch grown by max(P-1, 2) per side
    constexpr int GR = PM + 4;        // grey region edge for the gradient update
    __shared__ double s_c[4];
    __shared__ unsigned s_i[4];
    __shared__ int s_filled;
    __shared__ uint8_t s_px[PM * PM][4 + CRIM_NPL];   // source pixel: B,G,R, the planes, grey
    __shared__ uint8_t s_m[MR * MR];    // mask around the patch, updated as pixels are filled
    __shared__ uint8_t s_m0[MR * MR];   // the same before this iteration's fill
    __shared__ uint8_t s_g[GR * GR];
    __shared__ unsigned s_fill[(PM * PM + 31) / 32];   // bit e: patch element e was filled
    __shared__ unsigned long long s_rb[MR][2];         // per snapshot row: bit j = pixel j masked
    __shared__ unsigned long long s_rw[MR][2];         // OR of P consecutive rows
    __shared__ int s_ovf;                              // the tile list outgrew its capacity
    __shared__ unsigned s_tw[4];
    __shared__ int s_td[9];                            // change of the front count of the (up to 3 x 3) tiles around the patch
    __shared__ unsigned s_napp;                        // tiles appended to the list by this iteration

    // The kernel is a chain of dependent memory round trips, so independent loads are issued
    // together.  Load batch 1: loop state, descriptor, and the partial arg-min list (read up to its
    // capacity; entries beyond the count are ignored once the count has arrived).
    // The slots are visited in a compact order that skips the block columns the active tiling never
    // uses: compact index c is block row c / scols, block column c % scols.
    const SEntry* sl = p.slist + (size_t)b * p.snx * p.srows;
    const int scols = p.scols;
    auto slot_of = [&](int c) { const int r = c / scols; return imin(r, p.sny - 1) * p.snx + (c - r * scols); };
    constexpr int NSPEC = 4;
    SEntry spec[NSPEC];
#pragma unroll
    for (int k = 0; k < NSPEC; ++k) spec[k] = sl[slot_of(tid + 256 * k)];   // used only when in range
    const int live = st->live;
    // loop state is read once, up front: thread 0 rewrites it at the end of this kernel
    const int cur = st->cur;
    const int remaining0 = st->remaining, iter0 = st->iter;
    const unsigned tcount0 = st->tcount;
    const int tlsel = st->tlsel;
    const int ncx = d->ncx;
    // entries to reduce: the slots of the search blocks inside the candidate grid (a split search fills
    // them part by part, in this process or -- exchanged before this launch -- in others)
    const int grows = (d->ncy + SB_H - 1) / SB_H;
    const int gcols = search_cols(p, ncx);
    const int n = grows * scols;   // compact indices that can name a slot inside the candidate grid
    const int r0 = d->r0, c0 = d->c0, ph = d->ph, pw = d->pw;
    const double cnew = d->cnew;
    if (!live) return;
    // the (up to 3 x 3) 16 x 16 tiles that pixels within one pixel of the patch lie in, and their
    // front counts, requested early
    const int tty0 = imax(r0 - 1, 0) / CRIM_TILE, ttx0 = imax(c0 - 1, 0) / CRIM_TILE;
    int tf_old = 0;
    if (tid < 9) {
        const int ti = imin(tty0 + tid / 3, p.tny - 1), tj = imin(ttx0 + tid % 3, p.tnx - 1);
        tf_old = p.tfront[(size_t)b * p.tnx * p.tny + ti * p.tnx + tj];
    }
    STAMP(16);
    uint8_t* mask = p.mask + b * HW;
    uint8_t* bgr = p.bgr + b * HW * 3;
    uint8_t* grey = p.grey + b * HW;
    uint8_t* planes = p.planes + (size_t)b * CRIM_NPL * H * p.Wp;
    const size_t PS = (size_t)H * p.Wp;
    double* cnxt = p.conf + ((size_t)b * 2 + (cur ^ 1)) * HW;

    // Load batch 2 (needs only the descriptor): the mask around the patch, the grey values around it,
    // and the fresh confidence of the pixels within one pixel of the patch.
    const int mg = P - 1 > 2 ? P - 1 : 2;   // margin of the mask snapshot around the patch
    const int mr0 = r0 - mg, mc0 = c0 - mg;
    const int mrh = ph + 2 * mg, mrw = pw + 2 * mg;
    // Every load of this batch is unconditional, from a position clamped into the image, and the
    // out-of-image value is selected afterwards: the requests of a batch travel together.
    constexpr int NMB = (MR * MR + 255) / 256 < 8 ? (MR * MR + 255) / 256 : 8;   // mask loads per batch
    constexpr int NGB = (GR * GR + 255) / 256;
    constexpr int NRF = ((PM + 2) * (PM + 2) + 255) / 256;
    const int gr0 = r0 - 2, gc0 = c0 - 2, grh = ph + 4, grw = pw + 4;
    double rfv[NRF];   // confidence (fresh buffer) of this thread's pixels of the box grown by one
    uint8_t gm[NGB], gg[NGB];
    {
        const int cw = pw + 2;
#pragma unroll
        for (int k = 0; k < NRF; ++k) {
            const int e = tid + 256 * k;
            const int i = e / cw, j = e - i * cw;
            const int yy = imin(imax(r0 - 1 + i, 0), H - 1), xx = imin(imax(c0 - 1 + j, 0), W - 1);
            rfv[k] = cnxt[(size_t)yy * W + xx];
        }
#pragma unroll
        for (int k = 0; k < NGB; ++k) {
            const int e = tid + 256 * k;
            const int i = e / grw, j = e - i * grw;
            const size_t q = (size_t)imin(imax(gr0 + i, 0), H - 1) * W + imin(imax(gc0 + j, 0), W - 1);
            gm[k] = mask[q];
            gg[k] = grey[q];
        }
    }
    for (int e0 = tid; e0 < mrh * mrw; e0 += 256 * NMB) {
        uint8_t mvv[NMB];
#pragma unroll
        for (int k = 0; k < NMB; ++k) {
            const int e = e0 + 256 * k;
            const int i = e / mrw, j = e - i * mrw;
            mvv[k] = mask[(size_t)imin(imax(mr0 + i, 0), H - 1) * W + imin(imax(mc0 + j, 0), W - 1)];
        }
#pragma unroll
        for (int k = 0; k < NMB; ++k) {
            const int e = e0 + 256 * k;
            if (e >= mrh * mrw) continue;
            const int i = e / mrw, j = e - i * mrw;
            const int yy = mr0 + i, xx = mc0 + j;
            // beyond the image: a window reaching there is not a candidate
            const uint8_t v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? mvv[k] : (uint8_t)1;
            s_m[i * MR + j] = v;
            s_m0[i * MR + j] = v;
        }
    }
#pragma unroll
    for (int k = 0; k < NGB; ++k) {
        const int e = tid + 256 * k;
        if (e >= grh * grw) continue;
        const int i = e / grw, j = e - i * grw;
        const int yy = gr0 + i, xx = gc0 + j;
        uint8_t v = 0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = gm[k] ? (uint8_t)0 : gg[k];
        s_g[i * GR + j] = v;
    }
    {
        const int ch = ph + 2, cw = pw + 2;
#pragma unroll
        for (int k = 0; k < NRF; ++k) {
            const int e = tid + 256 * k;
            const int i = e / cw, j = e - i * cw;
            const int yy = r0 - 1 + i, xx = c0 - 1 + j;
            if (!(e < ch * cw && yy >= 0 && yy < H && xx >= 0 && xx < W)) rfv[k] = 0.0;
        }
    }

    // ---- arg-min over the per-block candidates: lowest cost, lowest row-major index (:264)
    double bc = INFINITY;
    unsigned bi = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < NSPEC; ++k) {
        const int i = tid + 256 * k;
        if (i < n && i % scols < gcols && better_cost(spec[k].cost, spec[k].idx, bc, bi)) {
            bc = spec[k].cost; bi = spec[k].idx;
        }
    }
    // what is left (large images only), four independent loads at a time
    for (int i0 = tid + 256 * NSPEC; i0 < n; i0 += 256 * NSPEC) {
        SEntry more[NSPEC];
#pragma unroll
        for (int k = 0; k < NSPEC; ++k) more[k] = sl[slot_of(imin(i0 + 256 * k, n - 1))];
#pragma unroll
        for (int k = 0; k < NSPEC; ++k) {
            const int i = i0 + 256 * k;
            if (i < n && i % scols < gcols && better_cost(more[k].cost, more[k].idx, bc, bi)) {
                bc = more[k].cost; bi = more[k].idx;
            }
        }
    }
    {
        int bi_ = (int)bi;
        wave_best_pair(bc, bi_, [](double ca, int ia, double cb, int ib) { return better_cost(ca, (unsigned)ia, cb, (unsigned)ib); });
        bi = (unsigned)bi_;
    }
    if ((tid & 63) == 0) { s_c[tid >> 6] = bc; s_i[tid >> 6] = bi; }
    if (tid == 0) { s_filled = 0; s_ovf = 0; s_napp = 0u; }
    if (tid < 9) s_td[tid] = 0;
    if (tid < (PM * PM + 31) / 32) s_fill[tid] = 0;
    __syncthreads();
    // every thread combines the four waves' results itself: one barrier instead of two
    bc = s_c[0]; bi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (better_cost(s_c[w], s_i[w], bc, bi)) { bc = s_c[w]; bi = s_i[w]; }
    // no admissible window: every cost is +inf and np.argmin returns index 0
    const unsigned src = (bi == 0xffffffffu) ? 0u : bi;
    const double src_cost = bc;
    STAMP(17);
    const int sy = (int)(src / (unsigned)ncx), sx = (int)(src % (unsigned)ncx);

    // Load batch 3 (needs the source position): the source pixels.
    for (int e = tid; e < ph * pw; e += 256) {
        const int i = e / pw, j = e - i * pw;
        const size_t s = (size_t)(sy + i) * W + sx + j;
        const size_t so = (size_t)(sy + i) * p.Wp + sx + j;
        uint8_t v[4 + CRIM_NPL];
        v[0] = bgr[3 * s]; v[1] = bgr[3 * s + 1]; v[2] = bgr[3 * s + 2];
#pragma unroll
        for (int k = 0; k < CRIM_NPL; ++k) v[3 + k] = planes[k * PS + so];
        v[3 + CRIM_NPL] = grey[s];
#pragma unroll
        for (int k = 0; k < 4 + CRIM_NPL; ++k) s_px[e][k] = v[k];
    }
    __syncthreads();
    STAMP(18);

    // ---- fill the masked pixels of the target patch (:274-308)
    // (the fill bits and the count go through wave ballots: same-address LDS atomics from every
    // lane would serialise)
    for (int e0 = 0; e0 < ph * pw; e0 += 256) {
        const int e = e0 + tid;
        const int i = e / pw, j = e - i * pw;
        const bool fill = e < ph * pw && s_m0[(i + mg) * MR + (j + mg)];
        if (fill) {
            const int yy = r0 + i, xx = c0 + j;
            const size_t q = (size_t)yy * W + xx;
            bgr[3 * q] = s_px[e][0]; bgr[3 * q + 1] = s_px[e][1]; bgr[3 * q + 2] = s_px[e][2];
            const size_t qo = (size_t)yy * p.Wp + xx;
            for (int k = 0; k < CRIM_NPL; ++k) planes[k * PS + qo] = s_px[e][3 + k];
            grey[q] = s_px[e][3 + CRIM_NPL];
            cnxt[q] = cnew;
            mask[q] = 0;
            s_m[(i + mg) * MR + (j + mg)] = 0;
            s_g[(i + 2) * GR + (j + 2)] = s_px[e][3 + CRIM_NPL];
        }
        const unsigned long long bal = __ballot(fill);
        if ((tid & 63) == 0 && bal != 0) {
            const int w = (e0 + tid) >> 5;   // this wave's 64 elements start on an even word
            constexpr int NW = (PM * PM + 31) / 32;
            if (w < NW) s_fill[w] = (unsigned)bal;
            if (w + 1 < NW) s_fill[w + 1] = (unsigned)(bal >> 32);
            atomicAdd(&s_filled, __popcll(bal));
        }
    }
    __syncthreads();
    STAMP(19);

    // ---- window-validity map: origins whose P x P window touches the patch
    uint8_t* inv = p.invalid + (size_t)b * H * p.Wi;
    const int oh = ph + P - 1, ow = pw + P - 1;   // origins r0-P+1 .. r1, c0-P+1 .. c1
    // Row bitmasks of the snapshot by wave ballots, then the OR over P consecutive rows, then one
    // shifted test per origin.
    const int og = mg - (P - 1);   // offset of the first origin inside the snapshot
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int row = wave; row < mrh; row += 4) {
            const unsigned long long lo = __ballot(lane < mrw && s_m[row * MR + lane]);
            const unsigned long long hi = __ballot(64 + lane < mrw && s_m[row * MR + 64 + lane]);
            if (lane == 0) { s_rb[row][0] = lo; s_rb[row][1] = hi; }
        }
    }
    __syncthreads();
    for (int i = tid; i < oh; i += 256) {
        unsigned long long lo = 0, hi = 0;
        for (int a = 0; a < P; ++a) { lo |= s_rb[og + i + a][0]; hi |= s_rb[og + i + a][1]; }
        s_rw[i][0] = lo; s_rw[i][1] = hi;
    }
    __syncthreads();
    // One wave per origin row, one lane per origin (ow <= 2P-1 <= 61): the byte map is stored, the
    // bitmap cleared with fire-and-forget atomics (a window only ever turns admissible).
    {
        const int lane = tid & 63, wave = tid >> 6;
        unsigned long long* ib = p.invbits + (size_t)b * H * p.Wb;
        for (int i = wave; i < oh; i += 4) {
            const int oy = r0 - (P - 1) + i, ox = c0 - (P - 1) + lane;
            const bool in = lane < ow && oy >= 0 && ox >= 0 && oy <= H - P && ox <= W - P;
            bool bad = true;
            if (in) {
                // bits [og + lane, og + lane + P) of the 128-bit row window
                const int sft = og + lane;
                const unsigned long long lo = s_rw[i][0], hi = s_rw[i][1];
                unsigned long long w;
                if (sft == 0) w = lo;
                else if (sft < 64) w = (lo >> sft) | (hi << (64 - sft));
                else w = hi >> (sft - 64);
                const unsigned long long pm = (1ull << P) - 1ull;
                bad = (w & pm) != 0;
                inv[(size_t)oy * p.Wi + ox] = (uint8_t)bad;
            }
            const unsigned long long good = __ballot(in && !bad);   // bit l: origin ox0 + l is admissible
            if (good != 0 && lane < 2) {
                // origins ox0 .. ox0+63 straddle at most two 64-bit words
                const int ox0 = c0 - (P - 1);
                const int wi = (ox0 >> 6) + lane;          // arithmetic shift: floor for negative ox0
                const int sh = ox0 & 63;
                unsigned long long clr;
                if (lane == 0) clr = good << sh;
                else clr = sh ? good >> (64 - sh) : 0ull;
                if (clr != 0 && wi >= 0 && wi < p.Wb) atomicAnd(ib + (size_t)oy * p.Wb + wi, ~clr);
            }
        }
    }
    // ---- per-tile front counts: only pixels within one pixel of the patch can join the front
    // (a filled pixel next to still-masked ones) or leave it (its masked diagonal neighbours
    // were all filled)
    {
        // Only this block changes its image's counts, so they need no global atomics: the changes are
        // summed per tile in LDS, and one thread per tile applies them to the count it requested with
        // the first batch of loads.  A tile enters the list when its count leaves zero.
        auto MN = [&](int y2, int x2) -> int {
            return (int)s_m[(reflect101(y2, H) - mr0) * MR + (reflect101(x2, W) - mc0)];
        };
        auto MO = [&](int y2, int x2) -> int {
            return (int)s_m0[(reflect101(y2, H) - mr0) * MR + (reflect101(x2, W) - mc0)];
        };
        const int fh = ph + 2, fw = pw + 2;
        for (int e = tid; e < fh * fw; e += 256) {
            const int i = e / fw, j = e - i * fw;
            const int yy = r0 - 1 + i, xx = c0 - 1 + j;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const bool fo = MO(yy, xx) == 0 &&
                            MO(yy - 1, xx - 1) + MO(yy - 1, xx + 1) + MO(yy + 1, xx - 1) + MO(yy + 1, xx + 1) > 0;
            const bool fn = MN(yy, xx) == 0 &&
                            MN(yy - 1, xx - 1) + MN(yy - 1, xx + 1) + MN(yy + 1, xx - 1) + MN(yy + 1, xx + 1) > 0;
            if (fo == fn) continue;
            atomicAdd(&s_td[(yy / CRIM_TILE - tty0) * 3 + (xx / CRIM_TILE - ttx0)], fn ? 1 : -1);
        }
    }
    __syncthreads();
    if (tid < 9) {
        const int ti = tid / 3, tj = tid - 3 * ti;
        const int delta = s_td[tid];
        if (delta != 0 && tty0 + ti < p.tny && ttx0 + tj < p.tnx) {
            int* tf = p.tfront + (size_t)b * p.tnx * p.tny;
            int* tl = p.tlist + ((size_t)b * 2 + tlsel) * p.tcap;
            const int tile = (tty0 + ti) * p.tnx + ttx0 + tj;
            const int now = tf_old + delta;
            __hip_atomic_store(&tf[tile], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tf_old == 0 && now > 0) {
                const unsigned k = tcount0 + atomicAdd(&s_napp, 1u);
                if (k < (unsigned)p.tcap) tl[k] = tile;
                else s_ovf = 1;   // dropped: the list is rebuilt from the counts below
            }
        }
    }
    // ---- confidence buffers: the buffer that was current during this iteration is stale on this
    // iteration's front and patch.  Front pixels that stay on the front are rewritten by the next
    // iteration; every other stale pixel lies within one pixel of the patch box and is refreshed
    // here (pixels that remain on the front are refreshed too, harmlessly).
    {
        double* cold = p.conf + ((size_t)b * 2 + cur) * HW;
        const int ch = ph + 2, cw = pw + 2;
#pragma unroll
        for (int k = 0; k < NRF; ++k) {
            const int e = tid + 256 * k;
            if (e >= ch * cw) continue;
            const int i = e / cw, j = e - i * cw;
            const int yy = r0 - 1 + i, xx = c0 - 1 + j;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const size_t q = (size_t)yy * W + xx;
            double v = rfv[k];
            const int pi = i - 1, pj = j - 1;
            if (pi >= 0 && pi < ph && pj >= 0 && pj < pw &&
                (s_fill[(pi * pw + pj) >> 5] & (1u << ((pi * pw + pj) & 31))))
                v = cnew;
            cold[q] = v;
        }
    }
    STAMP(21);
    // ---- gradient field: pixels within one pixel of the patch
    int* grad = p.grad + b * HW;
    const int uh = ph + 2, uw = pw + 2;
    for (int e = tid; e < uh * uw; e += 256) {
        const int i = e / uw, j = e - i * uw;
        const int yy = r0 - 1 + i, xx = c0 - 1 + j;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        auto G = [&](int y2, int x2) -> int { return (int)s_g[(y2 - gr0) * GR + (x2 - gc0)]; };
        int gy2, gx2;
        if (H == 1) gy2 = 0;
        else if (yy == 0) gy2 = 2 * (G(1, xx) - G(0, xx));
        else if (yy == H - 1) gy2 = 2 * (G(yy, xx) - G(yy - 1, xx));
        else gy2 = G(yy + 1, xx) - G(yy - 1, xx);
        if (W == 1) gx2 = 0;
        else if (xx == 0) gx2 = 2 * (G(yy, 1) - G(yy, 0));
        else if (xx == W - 1) gx2 = 2 * (G(yy, xx) - G(yy, xx - 1));
        else gx2 = G(yy, xx + 1) - G(yy, xx - 1);
        grad[(size_t)yy * W + xx] = (gy2 & 0xffff) | (int)((unsigned)gx2 << 16);
    }
    STAMP(22);
    // ---- tile list: entries are only ever appended between rebuilds (k_tiles runs in the long graphs
    // alone), so a list that would outgrow its capacity is rebuilt here, from the exact counts, in
    // row-major tile order.  This block is the only writer of its image's counts and list.
    __syncthreads();
    if (s_ovf) {
        const int lane = tid & 63, wv = tid >> 6;
        const int nt = p.tnx * p.tny;
        int* tf = p.tfront + (size_t)b * nt;
        int* tl = p.tlist + ((size_t)b * 2 + tlsel) * p.tcap;
        unsigned base = 0;
        for (int t0 = 0; t0 < nt; t0 += 256) {
            const int t = t0 + tid;
            // (the counts were last changed by this block's atomics: read them where those landed)
            const bool on = t < nt && __hip_atomic_load(&tf[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
            const unsigned long long bal = __ballot(on);
            if (lane == 0) s_tw[wv] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned off = base;
            for (int w = 0; w < wv; ++w) off += s_tw[w];
            if (on) tl[off + __popcll(bal & ((1ull << lane) - 1ull))] = t;
            base += s_tw[0] + s_tw[1] + s_tw[2] + s_tw[3];
            __syncthreads();
        }
        if (tid == 0) {
            st->tcount = base;
            st->rebuilds += 1;
        }
    } else if (tid == 0 && s_napp) {
        st->tcount = tcount0 + s_napp;
    }
    // ---- loop state and trace
    if (tid == 0) {
        const int nf = s_filled;
        const int rem = remaining0 - nf;
        const int it = iter0;
        if (p.trace && it < p.trace_cap) {
            crim_iter_rec rec;
            rec.tr = d->tr; rec.tc = d->tc; rec.sy = sy; rec.sx = sx;
            rec.cost = src_cost; rec.remaining = rem;
            p.trace[(size_t)b * p.trace_cap + it] = rec;
            if (p.stime) {
                long long* q = p.stime + ((size_t)b * p.trace_cap + it) * 3;
                q[0] = d->t_entry; q[1] = d->t_select; q[2] = t_entry;
            }
        }
        st->remaining = rem;
        st->iter = it + 1;
        st->cur = cur ^ 1;
        st->bound = 0x7fffffff;   // the next search starts without a pruning bound
        if (rem <= 0) st->live = 0;
        else if (nf == 0) { st->live = 0; st->stalled = 1; }
    }
    STAMP(23);
}

#include "crim_step.h"
