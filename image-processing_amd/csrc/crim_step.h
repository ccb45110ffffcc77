// k_step: everything of an iteration that follows the search, and the priority stage of the NEXT
// iteration, as ONE launch of independent blocks (small batches and single images, whose iterations are
// chains of dependent launches and memory round trips rather than work).
//
// What makes the blocks independent (criminisi_inpainter.py:51-83, :267-308):
//  * of the update, only the copied pixels (image, search planes, grey, gradient) depend on the source
//    window the search found; clearing the mask, the confidence of the filled pixels (:274-284), the
//    window-validity maps, the tile counts and the loop state follow from the TARGET alone, which the
//    search launch left in the descriptor together with the loop state it saw;
//  * the next iteration's confidence and priority of a front pixel read nothing this update writes
//    unless the pixel's patch reaches the filled patch grown by one pixel.  Tiles that can hold such a
//    pixel ("near": at most 3 x 3 of them, the ones the priority stage calls fresh) are evaluated by
//    waves that first work out the source window themselves and then stage the tile with the update
//    applied on the fly -- mask cleared inside the patch, the filled pixels' confidence, the gradient
//    recomputed within one pixel of the patch -- whether or not the blocks that write those fields to
//    memory have got there yet.  All other tiles ("far") are evaluated from memory as k_priority does.
//
// Block kinds of image b (p.kblocks blocks per image):
//   0          bookkeeping: mask, confidence buffers, validity maps, tile counts and list, loop state
//   1 .. 9     near tile k - 1, its four waves sharing the tile; block 9 also copies the pixels,
//              rewrites the gradient around the patch and records the iteration
//   10 ..      far tiles: wave w of these blocks walks entries w, w + p.nfar, ... of the tile list
// Every near block and every far wave owns one slot of the image's priority list, as the waves of
// k_priority do.
#pragma once

// Orders the LDS accesses of ONE wave (the waves of a k_step block work on different tiles and must not
// meet at block barriers): a wave's LDS instructions execute in order, so only the compiler has to be
// kept from moving accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS working set of one priority wave (the layout k_priority uses).
template <int PT>
struct PrioLds {
    static constexpr int RS = CRIM_TILE + PT - 1;       // edge of the staged region
    static constexpr int MS = CRIM_TILE + 2;            // edge of the staged mask
    static constexpr int NE = (PT * PT) / 8;
    double* s_cf;
    uint2* s_gk;
    uint8_t* s_list;
    uint8_t* s_mask;
    ushort2* s_off;
    __device__ __forceinline__ void carve(unsigned char* base) {
        s_cf = (double*)base;
        s_gk = (uint2*)(s_cf + RS * RS);
        s_list = (uint8_t*)(s_gk + RS * RS);
        s_mask = s_list + CRIM_TILE * CRIM_TILE;
        s_off = (ushort2*)(base + ((RS * RS * 16 + CRIM_TILE * CRIM_TILE + MS * MS + 3) & ~3));
    }
    // per-lane-of-group table of record offsets of the fast path (see k_priority)
    __device__ __forceinline__ void fill_offsets(int lane, int nthreads = 64) const {
        for (int t = lane; t < PG * NE; t += nthreads) {
            const int ll = t / NE, i = t - ll * NE;
            ushort2 o;
            {
                const int e = 8 * i + 2 * ll;
                const int row = e / PT, col = e - row * PT;
                o.x = (row * RS + col) * 8;
            }
            {
                const int e = 8 * i + 2 * ll + 1;
                const int row = e / PT, col = e - row * PT;
                o.y = (row * RS + col) * 8;
            }
            s_off[ll * NE + i] = o;
        }
    }
};

// The fields of one tile on their way from memory to LDS: requested by issue(), written by commit() (with
// the pending update applied when OVR), so that other work can pass in between.  NW waves share the tile
// (thread index t = 0 .. 64 NW - 1): a sweep of theirs covers 2 NW rows of up to 32 columns.
template <int PT, int NW = 1>
struct TileStage {
    using L = PrioLds<PT>;
    static constexpr int RPS = 2 * NW;                          // rows per sweep
    static constexpr int NRB = (L::RS + RPS - 1) / RPS;         // sweeps over the staged region
    static constexpr int MSW = (L::MS + RPS - 1) / RPS;         // sweeps over the staged mask
    static constexpr bool ROWS_FIT = L::RS % RPS == 0, MROWS_FIT = L::MS % RPS == 0;
    double cv[NRB];
    int gv[NRB];
    uint8_t mv[MSW];
    int ty0, tx0;

    __device__ __forceinline__ static int refl(int v, int n) {   // reflect101 of v in [-1, n]: -1 -> 1, n -> n - 2
        v = imax(v, -v);
        v = v >= n ? n - 2 : v;
        return imax(v, 0);
    }
    template <bool GRAD>
    __device__ __forceinline__ void issue(const Params& p, int tile, const double* conf, const int* grad,
                                          const uint8_t* mask, int lane) {
        const int H = p.H, W = p.W, half = p.half;
        const int tY = tile / p.tnx, tX = tile - tY * p.tnx;
        ty0 = tY * CRIM_TILE; tx0 = tX * CRIM_TILE;
        const unsigned mx = (unsigned)refl(tx0 - 1 + imin(lane & 31, L::MS - 1), W);
#pragma unroll
        for (int k = 0; k < MSW; ++k) {
            const int my = refl(ty0 - 1 + imin(RPS * k + (lane >> 5), L::MS - 1), H);
            mv[k] = mask[__umul24((unsigned)my, (unsigned)W) + mx];
        }
        const int xx = imin(imax(tx0 - half + (lane & 31), 0), W - 1);
#pragma unroll
        for (int k = 0; k < NRB; ++k) {
            const int row = imin((lane >> 5) + RPS * k, L::RS - 1);
            const int yy = imin(imax(ty0 - half + row, 0), H - 1);
            const unsigned idx = __umul24((unsigned)yy, (unsigned)W) + (unsigned)xx;
            cv[k] = *(const double*)((const char*)conf + (size_t)(idx * 8u));
            if (GRAD) gv[k] = *(const int*)((const char*)grad + (size_t)(idx * 4u));
        }
    }
};

// What an update that may not have reached memory yet changes in the fields a near tile stages.
struct PatchOverride {
    int r0, c0, ph, pw;       // the patch being filled
    double cnew;              // confidence its masked pixels receive (:274-284)
    const uint8_t* kflat;     // LDS: 1 = pixel known before the fill; row length = patch edge
    const int* ngrad;         // LDS: packed gradient after the fill over the patch grown by one pixel; row length patch edge + 2
};

template <int PT, bool GRAD, bool OVR, int NW = 1>
__device__ __forceinline__ void tile_commit(const Params& p, const TileStage<PT, NW>& st, const PrioLds<PT>& w,
                                            const PatchOverride& ov, int lane) {
    using L = PrioLds<PT>;
    using S = TileStage<PT, NW>;
    const int H = p.H, W = p.W, half = p.half;
    if ((lane & 31) < L::MS) {
        const int mx = S::refl(st.tx0 - 1 + imin(lane & 31, L::MS - 1), W);
#pragma unroll
        for (int k = 0; k < S::MSW; ++k) {
            uint8_t m = st.mv[k];
            const int mrow = S::RPS * k + (lane >> 5);
            if (OVR) {
                const int my = S::refl(st.ty0 - 1 + imin(mrow, L::MS - 1), H);
                if (my >= ov.r0 && my < ov.r0 + ov.ph && mx >= ov.c0 && mx < ov.c0 + ov.pw) m = 0;
            }
            if (S::MROWS_FIT || mrow < L::MS) w.s_mask[mrow * L::MS + (lane & 31)] = m;
        }
    }
    const int scol = lane & 31;
    const int xx = imin(imax(st.tx0 - half + scol, 0), W - 1);
    // the override tables are read for every row first (from clamped positions, all reads in flight
    // together), and applied afterwards
    uint8_t okf[OVR ? S::NRB : 1];
    if (OVR) {
        const int j = xx - ov.c0;
        const int jk = imin(imax(j, 0), ov.pw - 1);
#pragma unroll
        for (int k = 0; k < S::NRB; ++k) {
            const int yy = imin(imax(st.ty0 - half + imin((lane >> 5) + S::RPS * k, L::RS - 1), 0), H - 1);
            const int i = yy - ov.r0;
            okf[k] = ov.kflat[imin(imax(i, 0), ov.ph - 1) * PT + jk];
        }
    }
    static_assert(L::RS % 2 == 0, "with one wave a lane's rows (lane >> 5) + 2k stay inside the staged region");
    if (scol < L::RS) {   // one lane predicate for all rows
#pragma unroll
    for (int k = 0; k < S::NRB; ++k) {
        const int row = (lane >> 5) + S::RPS * k;
        if (S::ROWS_FIT || row < L::RS) {
            double c = st.cv[k];
            int g = GRAD ? st.gv[k] : 0;
            if (OVR) {
                const int yy = imin(imax(st.ty0 - half + row, 0), H - 1);
                const int i = yy - ov.r0, j = xx - ov.c0;
                if (i >= 0 && i < ov.ph && j >= 0 && j < ov.pw && okf[k] == 0) c = ov.cnew;
            }
            w.s_cf[row * L::RS + scol] = c;
            if (GRAD) {
                const int gy2 = (int)(short)(g & 0xffff), gx2 = g >> 16;
                w.s_gk[row * L::RS + scol] = make_uint2((uint32_t)(gy2 * gy2 + gx2 * gx2) << 10, (uint32_t)g);
            }
        }
    }
    }
}

// The staged gradient records of the pixels within one pixel of the patch, rewritten from the gradient
// after the fill (ov.ngrad), once that is known: tile_commit<.., true, true> stages the tile from memory
// with the mask and confidence overrides alone, so that it can run before the source window is known.
template <int PT>
__device__ __forceinline__ void tile_patch_gradient(const Params& p, int ty0, int tx0, const PrioLds<PT>& w,
                                                    const PatchOverride& ov, int lane, int nthreads = 64) {
    using L = PrioLds<PT>;
    const int H = p.H, W = p.W, half = p.half;
    const int bh = ov.ph + 2, bw = ov.pw + 2;
    // Staged record (row, col) holds the pixel at the CLAMPED position (ty0 - half + row, tx0 - half + col);
    // records beyond the image are never read, so only in-image pixels are visited.
    for (int e = lane; e < bh * bw; e += nthreads) {
        const int i = e / bw, j = e - i * bw;
        const int yy = ov.r0 - 1 + i, xx = ov.c0 - 1 + j;
        const int row = yy - (ty0 - half), col = xx - (tx0 - half);
        if (yy < 0 || yy >= H || xx < 0 || xx >= W || row < 0 || row >= L::RS || col < 0 || col >= L::RS) continue;
        const int g = ov.ngrad[i * (PT + 2) + j];
        const int gy2 = (int)(short)(g & 0xffff), gx2 = g >> 16;
        w.s_gk[row * L::RS + col] = make_uint2((uint32_t)(gy2 * gy2 + gx2 * gx2) << 10, (uint32_t)g);
    }
}

// Lists the front pixels of the staged tile (known pixels with a masked diagonal neighbour, :103);
// returns their number.
template <int PT>
__device__ __forceinline__ int tile_front_list(const Params& p, const PrioLds<PT>& w, int ty0, int tx0, int lane) {
    using L = PrioLds<PT>;
    int n = 0;
    for (int pass = 0; pass < CRIM_TILE * CRIM_TILE / 64; ++pass) {
        const int pi = pass * 64 + lane;
        const int py = pi >> 4, px = pi & 15;
        const uint8_t* m0 = w.s_mask + py * L::MS + px;
        const bool isf = (ty0 + py < p.H) & (tx0 + px < p.W) & (m0[L::MS + 1] == 0) &
                         ((m0[0] + m0[2] + m0[2 * L::MS] + m0[2 * L::MS + 2]) > 0);
        const unsigned long long bal = __ballot(isf);
        if (isf) w.s_list[n + __popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)pi;
        n += __popcll(bal);
    }
    return n;
}

// The same by the four waves of a block, each taking 64 of the tile's 256 pixels; s_cnt: 4 ints of LDS.
// Ends with the block in step (every thread returns the total).
template <int PT>
__device__ __forceinline__ int tile_front_list_block(const Params& p, const PrioLds<PT>& w, int ty0, int tx0, int tid,
                                                     int* s_cnt) {
    using L = PrioLds<PT>;
    const int lane = tid & 63, wv = tid >> 6;
    const int py = tid >> 4, px = tid & 15;
    const uint8_t* m0 = w.s_mask + py * L::MS + px;
    const bool isf = (ty0 + py < p.H) & (tx0 + px < p.W) & (m0[L::MS + 1] == 0) &
                     ((m0[0] + m0[2] + m0[2 * L::MS] + m0[2 * L::MS + 2]) > 0);
    const unsigned long long bal = __ballot(isf);
    if (lane == 0) s_cnt[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = s_cnt[k];
        before += k < wv ? c : 0;
        total += c;
    }
    if (isf) w.s_list[before + __popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)tid;
    __syncthreads();
    return total;
}

__device__ __forceinline__ void emit_priority(const Params& p, int b, int slot, double bestP, int bestI, int lane) {
    wave_best_pair(bestP, bestI, [](double pa, int ia, double pb, int ib) { return better_prio(pa, ia, pb, ib); });
    if (lane == 0) {
        PEntry e;
        e.prio = bestP; e.pos = bestI; e.pad = 0;
        p.plist[(size_t)b * p.pgrid + slot] = e;
    }
}

// Is the tile one whose front pixels may need a new data term after the patch (r0, c0, ph, pw) was
// filled?  (the test k_priority applies; the near waves of k_step cover exactly these tiles)
__device__ __forceinline__ bool tile_is_near(const Params& p, int tile, int r0, int c0, int ph, int pw) {
    const int tY = tile / p.tnx, tX = tile - tY * p.tnx;
    const int ty0 = tY * CRIM_TILE, tx0 = tX * CRIM_TILE;
    return ty0 <= r0 + ph + p.half && ty0 + CRIM_TILE - 1 >= r0 - 1 - p.half &&
           tx0 <= c0 + pw + p.half && tx0 + CRIM_TILE - 1 >= c0 - 1 - p.half;
}

// ---- far tiles --------------------------------------------------------------------------------
template <int PT>
__device__ __forceinline__ void step_far(const Params& p, const int b, const Desc* __restrict__ d, const int fblock,
                                         unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = 4 * fblock + wv;                 // this wave among the image's far waves
    const int slot = CRIM_NEAR_TILES + w;
    const size_t HW = (size_t)p.H * p.W;
    // the state the selection saw: the iteration being finished runs on buffer `cur`, the next one reads
    // cur ^ 1 (what the priority stage and this update write) and writes `cur`
    const int cur = d->cur;
    const int nt = imin((int)d->tcount, p.tcap);
    const int* tl = p.tlist + ((size_t)b * 2 + d->tlsel) * p.tcap;
    const int* tf = p.tfront + (size_t)b * p.tnx * p.tny;
    const int r0 = d->r0, c0 = d->c0, ph = d->ph, pw = d->pw;
    const double* cread = p.conf + ((size_t)b * 2 + (cur ^ 1)) * HW;
    double* cwrite = p.conf + ((size_t)b * 2 + cur) * HW;
    const uint8_t* mask = p.mask + b * HW;
    double* dterm = p.dterm + b * HW;
    double bestP = -1.0;
    int bestI = 0x7fffffff;
    if (w < nt) {
        PrioLds<PT> L;
        L.carve(smem + (size_t)wv * p.plds);
        L.fill_offsets(lane);
        const int grp = lane / PG, l = lane % PG;
        int vzero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));   // see k_priority: the next entry travels as a vector load
        int tile_next = tl[w];
        for (int ti = w; ti < nt; ti += p.nfar) {
            const int tile = tile_next;
            int tile_next_v = 0;
            const int tfv = tf[tile];
            if (ti + p.nfar < nt) tile_next_v = tl[ti + p.nfar + vzero];
            TileStage<PT> stg;
            stg.template issue<false>(p, tile, cread, nullptr, mask, lane);
            const bool skip = tile_is_near(p, tile, r0, c0, ph, pw) || tfv <= 0;   // wave-uniform
            if (!skip) {
                tile_commit<PT, false, false>(p, stg, L, PatchOverride{}, lane);
                wave_lds_sync();
                const int n = tile_front_list<PT>(p, L, stg.ty0, stg.tx0, lane);
                wave_lds_sync();
                prio_eval_tile<PT>(p, L.s_cf, L.s_gk, L.s_list, L.s_mask, L.s_off, PrioLds<PT>::RS, stg.ty0, stg.tx0, n,
                                   false, dterm, cwrite, grp, l, bestP, bestI);
                wave_lds_sync();   // the staging buffers are reused by the next tile
            }
            tile_next = __builtin_amdgcn_readfirstlane(tile_next_v);
        }
    }
    emit_priority(p, b, slot, bestP, bestI, lane);
}

// ---- near tiles, pixel copy --------------------------------------------------------------------
template <int PT>
__device__ __forceinline__ void step_near(const Params& p, const int b, const Desc* __restrict__ d, const int nblock,
                                          unsigned char* smem) {
    // One near tile per block, its four waves sharing the tile.  The block that also copies the pixels:
    // the last one, whose tile (the far corner of the 3 x 3) is the one least often in use.
    constexpr int COPY_BLOCK = CRIM_NEAR_TILES - 1;
    constexpr int NW = 4;
    __shared__ int s_cnt[4];
    __shared__ double s_bp[4];
    __shared__ int s_bi[4];
    constexpr int GR = PT + 4;       // grey region: the patch grown by two pixels
    constexpr int NG = PT + 2;       // gradient region: the patch grown by one pixel
    __shared__ double s_c[4];
    __shared__ unsigned s_i[4];
    __shared__ uint8_t s_px[PT * PT][4 + CRIM_NPL];   // source pixel: B,G,R, the planes, grey
    __shared__ uint8_t s_g[GR * GR];                  // grey after the fill, masked pixels as 0 (:171)
    __shared__ int s_ng[NG * NG];                     // packed gradient after the fill
    __shared__ uint8_t s_kf[PT * PT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = p.H, W = p.W, half = p.half;
    const size_t HW = (size_t)H * W;
    const long long t_entry = (nblock == COPY_BLOCK && tid == 0) ? (long long)wall_clock64() : 0;

    STAMP_AT(5, 40);
    // Load batch 1: the search's per-block results (as k_update reads them) and the descriptor.
    const SEntry* sl = p.slist + (size_t)b * p.snx * p.srows;
    const int scols = p.scols;
    auto slot_of = [&](int c) { const int r = c / scols; return imin(r, p.sny - 1) * p.snx + (c - r * scols); };
    constexpr int NSPEC = 4;
    SEntry spec[NSPEC];
#pragma unroll
    for (int k = 0; k < NSPEC; ++k) spec[k] = sl[slot_of(tid + 256 * k)];
    uint8_t kf = 1;
    if (tid < PT * PT) kf = d->kflat[tid];
    const int live = d->live;   // written by this iteration's search launch
    const int ncx = d->ncx;
    const int grows = (d->ncy + SB_H - 1) / SB_H;
    const int gcols = search_cols(p, ncx);
    const int n = grows * scols;
    const int r0 = d->r0, c0 = d->c0, ph = d->ph, pw = d->pw;
    const int cur = d->cur;
    // (everything the block needs from the descriptor is read here, in one batch of scalar loads: a read
    // further down would cost the chain a round trip of its own)
    const double cnew = d->cnew;
    const int d_tr = d->tr, d_tc = d->tc, d_iter = d->iter, d_rem = d->remaining - d->nfill;
    const long long d_tsel = d->t_select, d_tent = d->t_entry;
    if (!live) return;
    // the tile's LDS working set, and the offset table of the fast path, while the loads travel
    PrioLds<PT> L;
    L.carve(smem);
    L.fill_offsets(tid, 256);

    uint8_t* mask = p.mask + b * HW;
    uint8_t* bgr = p.bgr + b * HW * 3;
    uint8_t* grey = p.grey + b * HW;
    uint8_t* planes = p.planes + (size_t)b * CRIM_NPL * H * p.Wp;
    const size_t PS = (size_t)H * p.Wp;
    const double* cread = p.conf + ((size_t)b * 2 + (cur ^ 1)) * HW;
    double* cwrite = p.conf + ((size_t)b * 2 + cur) * HW;
    int* grad = p.grad + b * HW;

    // Load batch 2 (needs the descriptor): this wave's tile, and the grey values and the mask around the
    // patch.  Everything unconditional, from clamped positions.
    const int t = nblock;
    const int ny0 = imax(r0 - 1 - half, 0) / CRIM_TILE, nx0 = imax(c0 - 1 - half, 0) / CRIM_TILE;
    const int ny1 = imin(r0 + ph + half, H - 1) / CRIM_TILE, nx1 = imin(c0 + pw + half, W - 1) / CRIM_TILE;
    const int tY = ny0 + t / CRIM_NEAR_SPAN, tX = nx0 + t % CRIM_NEAR_SPAN;
    const bool active = tY <= ny1 && tX <= nx1;   // block-uniform
    if (!active && nblock != COPY_BLOCK) {
        // nothing of the next front can lie here: the slot is emptied and the block is done
        if (wv == 0) emit_priority(p, b, t, -1.0, 0x7fffffff, lane);
        return;
    }
    TileStage<PT, NW> stg;
    stg.template issue<true>(p, imin(tY, p.tny - 1) * p.tnx + imin(tX, p.tnx - 1), cread, grad, mask, tid);
    constexpr int NGB = (GR * GR + 255) / 256;
    const int gr0 = r0 - 2, gc0 = c0 - 2, grh = ph + 4, grw = pw + 4;
    uint8_t gm[NGB], gg[NGB];
#pragma unroll
    for (int k = 0; k < NGB; ++k) {
        const int e = tid + 256 * k;
        const int i = e / grw, j = e - i * grw;
        const size_t q = (size_t)imin(imax(gr0 + i, 0), H - 1) * W + imin(imax(gc0 + j, 0), W - 1);
        gm[k] = mask[q];
        gg[k] = grey[q];
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
    if (lane == 0) { s_c[wv] = bc; s_i[wv] = bi; }
    if (tid < PT * PT) s_kf[tid] = kf;
    __syncthreads();
    bc = s_c[0]; bi = s_i[0];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2)
        if (better_cost(s_c[w2], s_i[w2], bc, bi)) { bc = s_c[w2]; bi = s_i[w2]; }
    // no admissible window: every cost is +inf and np.argmin returns index 0
    const unsigned src = (bi == 0xffffffffu) ? 0u : bi;
    const int sy = (int)(src / (unsigned)ncx), sx = (int)(src % (unsigned)ncx);
    STAMP_AT(5, 41);

    // Load batch 3 (needs the source position): the source pixels, one per thread (the copying block: all
    // of a pixel; the others only need the grey value).  Requested now, used further down.
    static_assert(PT * PT <= 256, "one source pixel per thread");
    const bool has_src = tid < ph * pw;
    const int si = has_src ? tid / pw : 0, sj = has_src ? tid - si * pw : 0;
    uint8_t sv[4 + CRIM_NPL];
    {
        const size_t s = (size_t)(sy + si) * W + sx + sj;
        if (nblock == COPY_BLOCK) {
            const size_t so = (size_t)(sy + si) * p.Wp + sx + sj;
            sv[0] = bgr[3 * s]; sv[1] = bgr[3 * s + 1]; sv[2] = bgr[3 * s + 2];
#pragma unroll
            for (int k = 0; k < CRIM_NPL; ++k) sv[3 + k] = planes[k * PS + so];
        }
        sv[3 + CRIM_NPL] = grey[s];
    }
    // ---- while those travel: this wave's tile of the next iteration's priority stage is staged with
    // what of the update is known already (mask cleared inside the patch, confidence of the filled
    // pixels) and its front pixels are listed
    PatchOverride ov;
    ov.r0 = r0; ov.c0 = c0; ov.ph = ph; ov.pw = pw; ov.cnew = cnew; ov.kflat = s_kf; ov.ngrad = s_ng;
    int nf = 0;
    if (active) {   // (block-uniform: the barriers inside are reached by every thread)
        tile_commit<PT, true, true, NW>(p, stg, L, ov, tid);
        __syncthreads();
        nf = tile_front_list_block<PT>(p, L, stg.ty0, stg.tx0, tid, s_cnt);
    }
    STAMP_AT(5, 44);
    if (has_src) {
        if (nblock == COPY_BLOCK) {
#pragma unroll
            for (int k = 0; k < 4 + CRIM_NPL; ++k) s_px[tid][k] = sv[k];
        }
        if (s_kf[si * PT + sj] == 0) s_g[(si + 2) * GR + (sj + 2)] = sv[3 + CRIM_NPL];
    }
    // grey around the patch as the gradient sees it after the fill: 0 on pixels still masked (the patch
    // itself is cleared), the source's grey on filled pixels
#pragma unroll
    for (int k = 0; k < NGB; ++k) {
        const int e = tid + 256 * k;
        if (e >= grh * grw) continue;
        const int i = e / grw, j = e - i * grw;
        const int yy = gr0 + i, xx = gc0 + j;
        const int pi = i - 2, pj = j - 2;
        const bool inpatch = pi >= 0 && pi < ph && pj >= 0 && pj < pw;
        if (inpatch && s_kf[pi * PT + pj] == 0) continue;   // filled: written above, from the source
        uint8_t v = 0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = (gm[k] && !inpatch) ? (uint8_t)0 : gg[k];
        s_g[i * GR + j] = v;
    }
    __syncthreads();
    STAMP_AT(5, 42);
    // ---- gradient within one pixel of the patch (np.gradient, :174)
    const int uh = ph + 2, uw = pw + 2;
    for (int e = tid; e < uh * uw; e += 256) {
        const int i = e / uw, j = e - i * uw;
        const int yy = r0 - 1 + i, xx = c0 - 1 + j;
        int pk = 0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
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
            pk = (gy2 & 0xffff) | (int)((unsigned)gx2 << 16);
            if (nblock == COPY_BLOCK) grad[(size_t)yy * W + xx] = pk;
        }
        s_ng[i * NG + j] = pk;
    }
    if (nblock == COPY_BLOCK) {
        // ---- the copy itself (:287-296): masked pixels of the target patch take the source's
        for (int e = tid; e < ph * pw; e += 256) {
            const int i = e / pw, j = e - i * pw;
            if (s_kf[i * PT + j] != 0) continue;
            const int yy = r0 + i, xx = c0 + j;
            const size_t q = (size_t)yy * W + xx;
            bgr[3 * q] = s_px[e][0]; bgr[3 * q + 1] = s_px[e][1]; bgr[3 * q + 2] = s_px[e][2];
            const size_t qo = (size_t)yy * p.Wp + xx;
#pragma unroll
            for (int k = 0; k < CRIM_NPL; ++k) planes[k * PS + qo] = s_px[e][3 + k];
            grey[q] = s_px[e][3 + CRIM_NPL];
        }
        if (tid == 0) {
            const int it = d_iter;
            if (p.trace && it < p.trace_cap) {
                crim_iter_rec rec;
                rec.tr = d_tr; rec.tc = d_tc; rec.sy = sy; rec.sx = sx;
                rec.cost = bc; rec.remaining = d_rem;
                p.trace[(size_t)b * p.trace_cap + it] = rec;
                if (p.stime) {
                    long long* q = p.stime + ((size_t)b * p.trace_cap + it) * 3;
                    q[0] = d_tent; q[1] = d_tsel; q[2] = t_entry;
                }
            }
        }
    }
    __syncthreads();   // s_ng is complete
    STAMP_AT(5, 43);

    // ---- the staged gradient gets the update's part, then the front pixels are evaluated
    double bestP = -1.0;
    int bestI = 0x7fffffff;
    if (active) {
        tile_patch_gradient<PT>(p, stg.ty0, stg.tx0, L, ov, tid, 256);
        __syncthreads();
        STAMP_AT(5, 45);
        prio_eval_tile<PT>(p, L.s_cf, L.s_gk, L.s_list, L.s_mask, L.s_off, PrioLds<PT>::RS, stg.ty0, stg.tx0, nf, true,
                           p.dterm + b * HW, cwrite, tid / PG, tid % PG, bestP, bestI, 256 / PG);
    }
    STAMP_AT(5, 46);
    // the block's best: every wave's, then the first thread's pick
    wave_best_pair(bestP, bestI, [](double pa, int ia, double pb, int ib) { return better_prio(pa, ia, pb, ib); });
    if (lane == 0) { s_bp[wv] = bestP; s_bi[wv] = bestI; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (better_prio(s_bp[k], s_bi[k], bestP, bestI)) { bestP = s_bp[k]; bestI = s_bi[k]; }
        PEntry e;
        e.prio = bestP; e.pos = bestI; e.pad = 0;
        p.plist[(size_t)b * p.pgrid + t] = e;
    }
    STAMP_AT(5, 47);
}

// ---- bookkeeping -------------------------------------------------------------------------------
// k_update without the pixels: everything that follows from the target patch alone.
template <int PM>
__device__ __forceinline__ void step_book(const Params& p, const int b, const Desc* __restrict__ d) {
    ImgState* st = p.st + b;
    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, P = p.P;
    const size_t HW = (size_t)H * W;
    constexpr int MR = 3 * PM - 2 + 2;  // mask region edge: patch grown by max(P-1, 2) per side
    __shared__ int s_filled;
    __shared__ uint8_t s_m[MR * MR];    // mask around the patch after the fill
    __shared__ uint8_t s_m0[MR * MR];   // the same before it
    __shared__ unsigned s_fill[(PM * PM + 31) / 32];        // bit e: patch element e is filled
    __shared__ unsigned s_fnew[((PM + 2) * (PM + 2) + 31) / 32];   // bit e: element e of the patch grown by one is on the next front
    static_assert(MR <= 64, "a row of the mask snapshot fits one 64-bit word (patch edges up to 21)");
    __shared__ unsigned long long s_rb[MR];            // per snapshot row: bit j = pixel j masked
    __shared__ unsigned long long s_rw[MR];            // OR of P consecutive rows
    __shared__ int s_ovf;
    __shared__ unsigned s_tw[4];
    __shared__ int s_td[9];
    __shared__ unsigned s_napp;

    STAMP_AT(0, 48);
    // loop state: this block is its only reader and writer during the launch
    const int cur = st->cur;
    const int remaining0 = st->remaining, iter0 = st->iter;
    const unsigned tcount0 = st->tcount;
    const int tlsel = st->tlsel;
    const int r0 = d->r0, c0 = d->c0, ph = d->ph, pw = d->pw;
    const double cnew = d->cnew;
    const int tty0 = imax(r0 - 1, 0) / CRIM_TILE, ttx0 = imax(c0 - 1, 0) / CRIM_TILE;
    int tf_old = 0;
    if (tid < 9) {
        const int ti = imin(tty0 + tid / 3, p.tny - 1), tj = imin(ttx0 + tid % 3, p.tnx - 1);
        tf_old = p.tfront[(size_t)b * p.tnx * p.tny + ti * p.tnx + tj];
    }
    uint8_t* mask = p.mask + b * HW;
    double* cnxt = p.conf + ((size_t)b * 2 + (cur ^ 1)) * HW;

    const int mg = P - 1 > 2 ? P - 1 : 2;   // margin of the mask snapshot around the patch
    const int mr0 = r0 - mg, mc0 = c0 - mg;
    const int mrh = ph + 2 * mg, mrw = pw + 2 * mg;
    constexpr int NMB = (MR * MR + 255) / 256 < 8 ? (MR * MR + 255) / 256 : 8;
    constexpr int NRF = ((PM + 2) * (PM + 2) + 255) / 256;
    double rfv[NRF];   // confidence (fresh buffer) of this thread's pixels of the box grown by one
    {
        const int cw = pw + 2;
#pragma unroll
        for (int k = 0; k < NRF; ++k) {
            const int e = tid + 256 * k;
            const int i = e / cw, j = e - i * cw;
            const int yy = imin(imax(r0 - 1 + i, 0), H - 1), xx = imin(imax(c0 - 1 + j, 0), W - 1);
            rfv[k] = cnxt[(size_t)yy * W + xx];
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
            const uint8_t v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? mvv[k] : (uint8_t)1;
            const bool inpatch = i >= mg && i < mg + ph && j >= mg && j < mg + pw;
            s_m[i * MR + j] = inpatch ? (uint8_t)0 : v;
            s_m0[i * MR + j] = v;
        }
    }
    if (tid == 0) { s_filled = 0; s_ovf = 0; s_napp = 0u; }
    if (tid < 9) s_td[tid] = 0;
    if (tid < (PM * PM + 31) / 32) s_fill[tid] = 0;
    if (tid < ((PM + 2) * (PM + 2) + 31) / 32) s_fnew[tid] = 0;
    __syncthreads();
    STAMP_AT(0, 49);

    // ---- the masked pixels of the target patch: confidence (:274-284) and mask (:299-308)
    for (int e0 = 0; e0 < ph * pw; e0 += 256) {
        const int e = e0 + tid;
        const int i = e / pw, j = e - i * pw;
        const bool fill = e < ph * pw && s_m0[(i + mg) * MR + (j + mg)];
        if (fill) {
            const size_t q = (size_t)(r0 + i) * W + c0 + j;
            cnxt[q] = cnew;
            mask[q] = 0;
        }
        const unsigned long long bal = __ballot(fill);
        if ((tid & 63) == 0 && bal != 0) {
            const int w = (e0 + tid) >> 5;
            constexpr int NW = (PM * PM + 31) / 32;
            if (w < NW) s_fill[w] = (unsigned)bal;
            if (w + 1 < NW) s_fill[w + 1] = (unsigned)(bal >> 32);
            atomicAdd(&s_filled, __popcll(bal));
        }
    }

    STAMP_AT(0, 50);
    // ---- window-validity maps: origins whose P x P window touches the patch (as k_update)
    uint8_t* inv = p.invalid + (size_t)b * H * p.Wi;
    const int oh = ph + P - 1, ow = pw + P - 1;
    const int og = mg - (P - 1);
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int row = wave; row < mrh; row += 4) {
            const unsigned long long lo = __ballot(lane < mrw && s_m[row * MR + lane]);
            if (lane == 0) s_rb[row] = lo;
        }
    }
    __syncthreads();
    for (int i = tid; i < oh; i += 256) {
        unsigned long long lo = 0;
#pragma unroll
        for (int a = 0; a < PM; ++a) lo |= s_rb[og + i + a];   // P == PM in this kernel
        s_rw[i] = lo;
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        unsigned long long* ib = p.invbits + (size_t)b * H * p.Wb;
        for (int i = wave; i < oh; i += 4) {
            const int oy = r0 - (P - 1) + i, ox = c0 - (P - 1) + lane;
            const bool in = lane < ow && oy >= 0 && ox >= 0 && oy <= H - P && ox <= W - P;
            bool bad = true;
            if (in) {
                const int sft = og + lane;                 // < MR <= 64
                const unsigned long long pm = (1ull << P) - 1ull;
                bad = ((s_rw[i] >> sft) & pm) != 0;
                inv[(size_t)oy * p.Wi + ox] = (uint8_t)bad;
            }
            const unsigned long long good = __ballot(in && !bad);
            if (good != 0 && lane < 2) {
                const int ox0 = c0 - (P - 1);
                const int wi = (ox0 >> 6) + lane;
                const int sh = ox0 & 63;
                unsigned long long clr;
                if (lane == 0) clr = good << sh;
                else clr = sh ? good >> (64 - sh) : 0ull;
                if (clr != 0 && wi >= 0 && wi < p.Wb) atomicAnd(ib + (size_t)oy * p.Wb + wi, ~clr);
            }
        }
    }
    STAMP_AT(0, 51);
    // ---- per-tile front counts (pixels within one pixel of the patch), and which of those pixels are
    // on the next iteration's front
    {
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
            if (fn) atomicOr(&s_fnew[e >> 5], 1u << (e & 31));
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
                else s_ovf = 1;
            }
        }
    }
    STAMP_AT(0, 52);
    // ---- confidence buffers: the buffer that was current during this iteration is stale on this
    // iteration's front and patch.  The pixels of the NEXT front are being rewritten by this launch's
    // priority waves (they must not be touched here); every other stale pixel lies within one pixel of
    // the patch box and is refreshed from the fresh buffer.
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
            if (s_fnew[e >> 5] & (1u << (e & 31))) continue;
            double v = rfv[k];
            const int pi = i - 1, pj = j - 1;
            if (pi >= 0 && pi < ph && pj >= 0 && pj < pw &&
                (s_fill[(pi * pw + pj) >> 5] & (1u << ((pi * pw + pj) & 31))))
                v = cnew;
            cold[(size_t)yy * W + xx] = v;
        }
    }
    // ---- tile list: a list that would outgrow its capacity is rebuilt from the exact counts, in
    // row-major tile order, into the image's OTHER list buffer (the far waves of this launch are walking
    // the current one), which then becomes current.
    __syncthreads();
    if (s_ovf) {
        const int lane = tid & 63, wv = tid >> 6;
        const int nt = p.tnx * p.tny;
        int* tf = p.tfront + (size_t)b * nt;
        int* tl = p.tlist + ((size_t)b * 2 + (tlsel ^ 1)) * p.tcap;
        unsigned base = 0;
        for (int t0 = 0; t0 < nt; t0 += 256) {
            const int t = t0 + tid;
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
            st->tlsel = tlsel ^ 1;
            st->rebuilds += 1;
        }
    } else if (tid == 0 && s_napp) {
        st->tcount = tcount0 + s_napp;
    }
    if (tid == 0) {
        const int nf = s_filled;
        const int rem = remaining0 - nf;
        st->remaining = rem;
        st->iter = iter0 + 1;
        st->cur = cur ^ 1;
        st->bound = 0x7fffffff;
        if (rem <= 0) st->live = 0;
        else if (nf == 0) { st->live = 0; st->stalled = 1; }
    }
    STAMP_AT(0, 53);
}

template <int P>
__global__ __launch_bounds__(256) void k_step(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int img_, sub;
    xcd_decode(blockIdx.x, p.kblocks, p.B, &img_, &sub);
    const int b = p.b0 + img_;
    const Desc* __restrict__ d = p.desc + b;
    // (the near blocks request the search's results before they look at anything else)
    if (sub >= 1 && sub <= CRIM_NEAR_TILES) {
        step_near<P>(p, b, d, sub - 1, smem);
        return;
    }
    if (!d->live) return;   // written by this iteration's search launch
    if (sub == 0) step_book<P>(p, b, d);
    else step_far<P>(p, b, d, sub - 1 - CRIM_NEAR_TILES, smem);
}
