/*
 * Plain-C restatement of the reference's Criminisi fill loop.
 *
 * TEST INFRASTRUCTURE ONLY: this is the checker for the HIP engine and the "port" CPU baseline
 * timed by bench.py.  It is never linked into, loaded by, or called from the shipped library.
 *
 * PARITY UNPINNED: the reference (/root/reference/src/criminisi_inpainter.py) needs OpenCV,
 * which is absent from the build container, and holds no fixtures.  This file follows the
 * reference's loop and keeps its cost structure (every stage recomputed over the full frame on
 * every iteration); OpenCV primitives are restated from their published algorithms.  It is
 * cross-checked against oracle/criminisi_np.py, which calls the real NumPy functions the
 * reference calls.
 *
 * Reference map (criminisi_inpainter.py): state :18-38, loop :51-83, front :103, confidence
 * :114-127, normal :138-164, gradient :166-214, priority :112, target :216-218, search
 * :220-265, update :267-308, patch geometry :310-323.
 *
 * The search uses the direct form sum k*(L-T)^2 - sum k*T^2, which equals the reference's
 * decomposed xcorr(I^2,k) - 2*xcorr(I,k*T) exactly in integers.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t tr, tc;      /* target pixel */
    int32_t sy, sx;      /* source window origin */
    double cost;         /* fp64 cost of the chosen window */
    int64_t remaining;   /* masked pixels left after the update */
} crim_oracle_rec;

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

/* NumPy's pairwise summation of a flattened double array (verified against np.sum on
 * non-contiguous 2-D slices): <8 sequential, <=128 eight interleaved accumulators, else split. */
static double pairwise_sum(const double *a, int n) {
    if (n < 8) {
        double r = -0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

static void bgr2lab(const uint8_t *bgr, uint8_t *lab, long npix, const uint16_t *gam,
                    const uint16_t *cbrt_tab, const int32_t *c) {
    for (long i = 0; i < npix; i++) {
        int B = gam[bgr[3 * i]], G = gam[bgr[3 * i + 1]], R = gam[bgr[3 * i + 2]];
        int fX = cbrt_tab[(B * c[0] + G * c[1] + R * c[2] + 2048) >> 12];
        int fY = cbrt_tab[(B * c[3] + G * c[4] + R * c[5] + 2048) >> 12];
        int fZ = cbrt_tab[(B * c[6] + G * c[7] + R * c[8] + 2048) >> 12];
        int L = (296 * fY - 1336934 + 16384) >> 15;
        int a = (500 * (fX - fY) + 128 * 32768 + 16384) >> 15;
        int b = (200 * (fY - fZ) + 128 * 32768 + 16384) >> 15;
        lab[3 * i] = (uint8_t)(L < 0 ? 0 : L > 255 ? 255 : L);
        lab[3 * i + 1] = (uint8_t)(a < 0 ? 0 : a > 255 ? 255 : a);
        lab[3 * i + 2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
    }
}

/* _find_source_patch (:220-265) on a Lab frame: first arg-min, row-major, of
 * float64(S) + distance over the windows free of masked pixels; (0,0) with cost +inf when there
 * is none.  sat is scratch for (H+1)*(W+1) ints. */
static void search(const uint8_t *lab, const uint8_t *msk, int32_t *sat, int H, int W, int r0,
                   int r1, int c0, int c1, int *sy_out, int *sx_out, double *cost_out) {
    const int ph = r1 - r0 + 1, pw = c1 - c0 + 1;
    /* integral image of the mask for the window test (:250-255) */
    const int SW = W + 1;
    for (int x = 0; x <= W; x++) sat[x] = 0;
    for (int y = 0; y < H; y++) {
        int run = 0;
        sat[(y + 1) * SW] = 0;
        for (int x = 0; x < W; x++) {
            run += msk[y * W + x];
            sat[(y + 1) * SW + x + 1] = sat[y * SW + x + 1] + run;
        }
    }
    long tconst = 0;
    for (int i = 0; i < ph; i++)
        for (int j = 0; j < pw; j++)
            if (!msk[(r0 + i) * W + c0 + j])
                for (int ch = 0; ch < 3; ch++) {
                    long t = lab[((r0 + i) * W + c0 + j) * 3 + ch];
                    tconst += t * t;
                }
    const double ty = (r0 + r1) / 2.0, tx = (c0 + c1) / 2.0;
    double best_c = INFINITY;
    int sy = 0, sx = 0;
    for (int y = 0; y <= H - ph; y++)
        for (int x = 0; x <= W - pw; x++) {
            int msum = sat[(y + ph) * SW + x + pw] - sat[y * SW + x + pw] -
                       sat[(y + ph) * SW + x] + sat[y * SW + x];
            if (msum > 0) continue;
            long s = 0;
            for (int i = 0; i < ph; i++) {
                const uint8_t *lrow = lab + ((long)(y + i) * W + x) * 3;
                const uint8_t *trow = lab + ((long)(r0 + i) * W + c0) * 3;
                const uint8_t *mrow = msk + (long)(r0 + i) * W + c0;
                for (int j = 0; j < pw; j++) {
                    if (mrow[j]) continue;
                    int d0 = lrow[3 * j] - trow[3 * j];
                    int d1 = lrow[3 * j + 1] - trow[3 * j + 1];
                    int d2 = lrow[3 * j + 2] - trow[3 * j + 2];
                    s += d0 * d0 + d1 * d1 + d2 * d2;
                }
            }
            double dyy = ((double)y + ph / 2.0) - ty, dxx = ((double)x + pw / 2.0) - tx;
            double cost = (double)(s - tconst) + sqrt(dyy * dyy + dxx * dxx);
            if (cost < best_c) { best_c = cost; sy = y; sx = x; }
        }
    *sy_out = sy; *sx_out = sx; *cost_out = best_c;
}

/* The search alone, for a given image, mask and target pixel (used to check sampled iterations
 * of runs too long to repeat on the CPU).  Returns 0, or -1 on allocation failure. */
int crim_oracle_search(const uint8_t *image_bgr, const uint8_t *mask, int H, int W, int p, int tr,
                       int tc, const uint16_t *gamma_tab, const uint16_t *cbrt_tab,
                       const int32_t *coeffs, int32_t *sy, int32_t *sx, double *cost) {
    const long np_ = (long)H * W;
    const int half = (p - 1) / 2;
    uint8_t *lab = malloc(np_ * 3), *msk = malloc(np_);
    int32_t *sat = malloc((np_ + H + W + 1) * sizeof(int32_t));
    if (!lab || !msk || !sat) { free(lab); free(msk); free(sat); return -1; }
    for (long i = 0; i < np_; i++) msk[i] = mask[i] ? 1 : 0;
    bgr2lab(image_bgr, lab, np_, gamma_tab, cbrt_tab, coeffs);
    int r0 = tr - half < 0 ? 0 : tr - half, r1 = tr + half > H - 1 ? H - 1 : tr + half;
    int c0 = tc - half < 0 ? 0 : tc - half, c1 = tc + half > W - 1 ? W - 1 : tc + half;
    int y, x;
    search(lab, msk, sat, H, W, r0, r1, c0, c1, &y, &x, cost);
    *sy = y; *sx = x;
    free(lab); free(msk); free(sat);
    return 0;
}

/*
 * Runs the loop until the mask is empty or max_iters iterations were done (max_iters <= 0: no
 * limit).  image/mask are not modified.  Returns the number of iterations executed, or -1 on
 * allocation failure.  trace may be NULL; at most trace_cap records are written.
 */
long crim_oracle_run(const uint8_t *image_bgr, const uint8_t *mask_in, int H, int W, int p,
                     const uint16_t *gamma_tab, const uint16_t *cbrt_tab, const int32_t *coeffs,
                     long max_iters, uint8_t *out_bgr, double *out_conf, uint8_t *out_mask,
                     crim_oracle_rec *trace, long trace_cap) {
    const long np_ = (long)H * W;
    const int half = (p - 1) / 2;
    uint8_t *img = malloc(np_ * 3), *msk = malloc(np_), *front = malloc(np_);
    uint8_t *lab = malloc(np_ * 3), *grey = malloc(np_);
    double *conf = malloc(np_ * sizeof(double)), *nconf = malloc(np_ * sizeof(double));
    double *gy = malloc(np_ * sizeof(double)), *gx = malloc(np_ * sizeof(double));
    double *gm = malloc(np_ * sizeof(double));
    int32_t *sat = malloc((np_ + H + W + 1) * sizeof(int32_t));
    double *pbuf = malloc((size_t)p * p * sizeof(double));
    uint8_t *srcbuf = malloc((size_t)p * p * 3);
    if (!img || !msk || !front || !lab || !grey || !conf || !nconf || !gy || !gx || !gm || !sat ||
        !pbuf || !srcbuf)
        return -1;
    memcpy(img, image_bgr, np_ * 3);
    long remaining = 0;
    for (long i = 0; i < np_; i++) {
        msk[i] = mask_in[i] ? 1 : 0;
        conf[i] = 1.0 - (double)msk[i];
        remaining += msk[i];
    }
    long it = 0;
    for (;;) {
        /* ---- front (:103): sat_u8(2*diagonals - 8*centre), BORDER_REFLECT_101 ---- */
        for (int y = 0; y < H; y++) {
            int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
            for (int x = 0; x < W; x++) {
                int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
                int v = 2 * (msk[ym * W + xm] + msk[ym * W + xp] + msk[yp * W + xm] +
                             msk[yp * W + xp]) - 8 * msk[y * W + x];
                front[y * W + x] = (uint8_t)(v < 0 ? 0 : v);
            }
        }
        /* ---- confidence (:114-127), Jacobi: reads old, writes copy ---- */
        memcpy(nconf, conf, np_ * sizeof(double));
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                if (!front[y * W + x]) continue;
                int r0 = y - half < 0 ? 0 : y - half, r1 = y + half > H - 1 ? H - 1 : y + half;
                int c0 = x - half < 0 ? 0 : x - half, c1 = x + half > W - 1 ? W - 1 : x + half;
                int n = 0;
                for (int r = r0; r <= r1; r++)
                    for (int c = c0; c <= c1; c++) pbuf[n++] = conf[r * W + c];
                nconf[y * W + x] = pairwise_sum(pbuf, n) / (double)(p * p);
            }
        { double *t = conf; conf = nconf; nconf = t; }
        /* ---- gradient field (:170-176): grey with the hole zeroed, np.gradient ---- */
        for (long i = 0; i < np_; i++) {
            int g = (img[3 * i] * 3735 + img[3 * i + 1] * 19235 + img[3 * i + 2] * 9798 + 16384) >> 15;
            grey[i] = msk[i] ? 0 : (uint8_t)g;
        }
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double dy, dx;
                if (H == 1) dy = 0.0;
                else if (y == 0) dy = (double)grey[W + x] - (double)grey[x];
                else if (y == H - 1) dy = (double)grey[y * W + x] - (double)grey[(y - 1) * W + x];
                else dy = ((double)grey[(y + 1) * W + x] - (double)grey[(y - 1) * W + x]) / 2.0;
                if (W == 1) dx = 0.0;
                else if (x == 0) dx = (double)grey[y * W + 1] - (double)grey[y * W];
                else if (x == W - 1) dx = (double)grey[y * W + x] - (double)grey[y * W + x - 1];
                else dx = ((double)grey[y * W + x + 1] - (double)grey[y * W + x - 1]) / 2.0;
                gy[y * W + x] = dy;
                gx[y * W + x] = dx;
                gm[y * W + x] = sqrt(dy * dy + dx * dx);
            }
        /* ---- data, priority, target (:112,:129-164,:183-218) ---- */
        double best_p = 0.0;
        long best_i = 0;
        int have = 0;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double pr = 0.0;
                if (front[y * W + x]) {
                    int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
                    int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
                    double a = msk[ym * W + xm], b = msk[ym * W + x], c = msk[ym * W + xp];
                    double d = msk[y * W + xm], f = msk[y * W + xp];
                    double g = msk[yp * W + xm], h = msk[yp * W + x], k = msk[yp * W + xp];
                    double nx = .25 * a - .25 * c + .5 * d - .5 * f + .25 * g - .25 * k;
                    double ny = -.25 * a - .5 * b - .25 * c + .25 * g + .5 * h + .25 * k;
                    double nrm = sqrt(ny * ny + nx * nx);
                    if (nrm == 0) nrm = 1;
                    double ux = nx / nrm, uy = ny / nrm;
                    int r0 = y - half < 0 ? 0 : y - half, r1 = y + half > H - 1 ? H - 1 : y + half;
                    int c0 = x - half < 0 ? 0 : x - half, c1 = x + half > W - 1 ? W - 1 : x + half;
                    double mv = -1.0;
                    long mi = 0;
                    for (int r = r0; r <= r1; r++)
                        for (int cc = c0; cc <= c1; cc++)
                            if (gm[r * W + cc] > mv) { mv = gm[r * W + cc]; mi = (long)r * W + cc; }
                    double t0 = ux * gy[mi], t1 = uy * gx[mi];
                    double data = sqrt(t0 * t0 + t1 * t1) + 0.001;
                    pr = conf[y * W + x] * data * (double)front[y * W + x];
                }
                if (!have || pr > best_p) { best_p = pr; best_i = (long)y * W + x; have = 1; }
            }
        int tr = (int)(best_i / W), tc = (int)(best_i % W);
        /* ---- search (:220-265) ---- */
        int r0 = tr - half < 0 ? 0 : tr - half, r1 = tr + half > H - 1 ? H - 1 : tr + half;
        int c0 = tc - half < 0 ? 0 : tc - half, c1 = tc + half > W - 1 ? W - 1 : tc + half;
        int ph = r1 - r0 + 1, pw = c1 - c0 + 1;
        bgr2lab(img, lab, np_, gamma_tab, cbrt_tab, coeffs);
        double best_c;
        int sy, sx;
        search(lab, msk, sat, H, W, r0, r1, c0, c1, &sy, &sx, &best_c);
        /* ---- update (:267-308): the reference builds the whole new patch before writing it,
         * so source pixels are read before any target pixel changes ---- */
        double pconf = conf[tr * W + tc];
        for (int i = 0; i < ph; i++)
            for (int j = 0; j < pw; j++) {
                long s = (long)(sy + i) * W + sx + j;
                srcbuf[(i * pw + j) * 3] = img[3 * s];
                srcbuf[(i * pw + j) * 3 + 1] = img[3 * s + 1];
                srcbuf[(i * pw + j) * 3 + 2] = img[3 * s + 2];
            }
        for (int i = 0; i < ph; i++)
            for (int j = 0; j < pw; j++) {
                long q = (long)(r0 + i) * W + c0 + j;
                if (!msk[q]) continue;
                conf[q] = pconf;
                img[3 * q] = srcbuf[(i * pw + j) * 3];
                img[3 * q + 1] = srcbuf[(i * pw + j) * 3 + 1];
                img[3 * q + 2] = srcbuf[(i * pw + j) * 3 + 2];
                msk[q] = 0;
                remaining--;
            }
        if (trace && it < trace_cap) {
            trace[it].tr = tr; trace[it].tc = tc; trace[it].sy = sy; trace[it].sx = sx;
            trace[it].cost = best_c; trace[it].remaining = remaining;
        }
        it++;
        if (remaining == 0) break;
        if (max_iters > 0 && it >= max_iters) break;
    }
    if (out_bgr) memcpy(out_bgr, img, np_ * 3);
    if (out_conf) memcpy(out_conf, conf, np_ * sizeof(double));
    if (out_mask) memcpy(out_mask, msk, np_);
    free(img); free(msk); free(front); free(lab); free(grey); free(conf); free(nconf);
    free(gy); free(gx); free(gm); free(sat); free(pbuf); free(srcbuf);
    return it;
}
