// Device/host shared data layout of the Criminisi engine (gfx950).
#pragma once
#include <stdint.h>

#include "../../include/criminisi_hip.h"

#define CRIM_TILE 16          // edge of a priority-scan tile (pixels)
#define CRIM_PMAX CRIM_MAX_PATCH
#define CRIM_NDMAX ((CRIM_PMAX + 6) / 4)
#define CRIM_NPL 7            // search byte planes per image
// 16x16 tiles that can hold a front pixel whose data term an iteration changes: the filled patch grown
// by the patch half-width plus one spans 2 P + 1 <= 33 pixels for patch edges up to 16, i.e. at most
// 3 tiles each way
#define CRIM_NEAR_SPAN 3
#define CRIM_NEAR_TILES (CRIM_NEAR_SPAN * CRIM_NEAR_SPAN)

// Matrix-core search tiling: a 256-thread block scores MS_W x MS_ROWS window origins; each of its
// 4 waves owns MS_R consecutive origin rows.  Patch edges up to MS_PMAX.
// entries of the Lab conversion's look-up tables (gamma per 8-bit value; cube root per 12-bit linear value)
#define CRIM_GAMMA_N 256
#define CRIM_CBRT_N 3072
#define MS_W 256
#define MS_R 8
#define MS_ROWS (4 * MS_R)
#define MS_PITCH (MS_W + 16)
#define MS_PMAX 15
#define MS_NSMAX ((MS_PMAX + 1) / 2)

// Search-kernel tiling: a 256-thread block scores SB_TXG*4 x SB_TYT*SB_R candidate origins.
#define SB_TXG 32   // groups of 4 consecutive x origins per block row
#define SB_TYT 8    // thread rows per block
#define SB_R 4      // origin rows per thread
#define SB_W (SB_TXG * 4)
#define SB_H (SB_TYT * SB_R)

// Per-image loop state, advanced on the device so that a replayed hipGraph needs no host input.
struct ImgState {
    int iter;        // iterations executed so far
    int live;        // 1 while the next iteration must run
    int cur;         // which of the two confidence buffers is current
    int remaining;   // masked pixels left
    unsigned tcount; // entries in the priority-stage tile list
    int chroma;      // some pixel of the image has Lab a or b different from 128
    unsigned pcount; // entries in this image's priority partial list
    unsigned scount; // entries in this image's search partial list
    int stalled;     // an iteration filled nothing while pixels remained
    int bound;       // running upper bound on floor(best cost) for search pruning
    int rebuilds;    // times k_update rebuilt a tile list that had outgrown its capacity
    int tlsel;       // which of the image's two tile-list buffers is current
    int pad[4];
};

// Best front pixel of one priority tile.
struct PEntry {
    double prio;
    int pos;
    int pad;
};

// Best candidate of one search block.
struct SEntry {
    double cost;
    unsigned idx;
    unsigned pad;
};

// Target-patch descriptor, rebuilt every iteration by the select kernel.
struct Desc {
    int live;
    int tr, tc;          // target pixel
    int r0, c0, ph, pw;  // target patch (clipped to the image)
    int clipped;         // ph or pw smaller than the full patch edge
    int ncy, ncx;        // candidate grid: H-ph+1, W-pw+1
    int soff;            // S_reference = S_centred + soff
    int soff2;           // S_reference = soff2 - (256*sum k*h + 2*cross + sum k*l), digits of 24576 - Q
    int cur;             // loop state as the selection saw it: current confidence buffer, ...
    int rowmask;         // bit i set: template row i has at least one known pixel
    int chmask;          // bit c set: k * (Lab_c - 128) is nonzero somewhere in the template
    int nfill;           // masked pixels of the target patch: what this iteration fills
    double cnew;         // updated confidence at the target pixel
    double ty, tx;       // patch centre as the reference computes it
    int iter, remaining; // ... iterations done and masked pixels left before this iteration,
    int tlsel;           // ... current tile-list buffer,
    unsigned tcount;     // ... and its length
    long long t_select;  // 100 MHz wall clock when the selection finished (the search starts here)
    long long t_entry;   // ... and when the launch that selects started
    // Templates as byte planes, zero padded: kflat = 1 - mask, tflat = k * (Lab - 128).
    uint8_t kflat[CRIM_PMAX * CRIM_PMAX];
    int8_t tflat[3][CRIM_PMAX * CRIM_PMAX];
    // Pre-shifted dword rows for the dot4 kernels: [type][row][shift][dword], type 0..2 =
    // k*T per Lab channel (signed), 3 = k, 4 = 255*k (unsigned).  Dimensions are the
    // engine's P and ND (compact, not CRIM_PMAX).
    uint32_t tmpl[5 * CRIM_PMAX * 4 * CRIM_NDMAX];
    // Template rows for the matrix-core search, each zero padded to 16 bytes: [type][row], type
    // 0..2 = k*T per Lab channel (signed), 3 = k.  Only written for patch edges <= MS_PMAX.
    __attribute__((aligned(16))) uint8_t zrows[4][MS_PMAX + 1][16];
};

struct Params {
    int B, H, W;  // B: images covered by a launch with these parameters
    int b0;       // first image covered (iteration kernels); upload-time kernels take `first`
    int P;        // patch edge 2*((patch_size-1)/2)+1
    int half;
    int psq;      // patch_size squared: the confidence divisor
    int ND;       // dwords per shifted template row
    int Wp;       // pitch of the search byte planes
    int Wi;       // pitch of the window-validity map
    int Wb;       // 64-bit words per row of the window-validity bitmap
    int tnx, tny; // 16x16 tile grid over the whole image (upload-time kernels)
    int snx, sny; // search block grid
    int srows;    // rows of the per-image slot array (sny rounded up so that equal shares of a split search fit)
    int sy0;      // first block row a search launch covers (split search; 0 otherwise)
    int dmax;     // upper bound of the distance term of the search cost: ceil(sqrt(H^2 + W^2)) + 1
    int mgx, mrows; // matrix-core search launch: blocks per row of tiles, rows of tiles covered
    int mtiles, mchunks; // pipelined matrix-core search: rows of tiles per block, blocks per column of tiles
    int pipe_force;   // host side: > 0 fixes the rows of tiles per block (experiments)
    int pipe_t, pipe_min_blocks; // host side: most rows of tiles per block (0 = plain kernel), launch size below which fewer are used
    int pgrid;    // priority slots per image: one per wave of the priority stage
    int fused;    // host side: an iteration is the search launch (which also selects) and ONE launch that
                  // updates the image and evaluates the next iteration's priorities (k_step)
    int nfar;     // k_step: waves per image that walk the tile list (slots CRIM_NEAR_TILES ..)
    int plds;     // bytes of LDS one priority wave works in
    int kblocks;  // k_step: blocks per image (1 bookkeeping + 9 near + nfar / 4 far)
    int generic;  // use the unspecialised search kernel
    int dot4;     // use the vector dot-product search kernel instead of the matrix-core one
    int chroma;   // some uploaded image is not grey: the matrix-core search stages all five planes
    long trace_cap;
    uint8_t* bgr;      // [B][H][W][3]
    uint8_t* planes;   // [B][CRIM_NPL][H][Wp]  L-128, a-128, b-128 (int8), Q%255, Q/255 (uint8),
                       // and the balanced digits l, h (int8) of 24576 - Q = 256*h + l
    uint8_t* mask;     // [B][H][W]
    uint8_t* grey;     // [B][H][W]
    int* grad;         // [B][H][W] packed (int16 2*gy, int16 2*gx)
    double* conf;      // [B][2][H][W]
    uint8_t* invalid;  // [B][H][Wi] 1: the P x P window at this origin holds a masked pixel
    unsigned long long* invbits;  // [B][H][Wb] the same map, one bit per origin (bit x%64 of word x/64)
    ImgState* st;
    Desc* desc;
    int tcap;          // capacity of one image's tile list (twice the tile count)
    int* tfront;       // [B][tny*tnx] front pixels per 16x16 tile
    int* tlist;        // [B][2][tcap] tiles the priority stage visits (two buffers: a rebuild that runs beside
                       // readers of the list writes the other one, ImgState::tlsel names the current one)
    PEntry* plist;     // [B][pgrid]
    SEntry* slist;     // [B][snx*srows] one slot per search block (row-major block rows)
    crim_iter_rec* trace;  // [B][trace_cap]
    long long* stime;      // [B][trace_cap][3] 100 MHz wall clock per iteration: start of the selecting launch, end of the
                           // selection, start of the update
    long long* dbg;        // diagnostic stamps (only written by -DCRIM_STAMPS builds)
    const uint16_t* gamma_tab;
    const uint16_t* cbrt_tab;
    int coef[9];
    int scols;         // most block columns of the active search tiling that can lie inside a candidate grid
    int selsearch;     // host side: the pipelined search also picks the target and builds the template (no k_select launch)
    double* dterm;     // [B][H][W] data term of a front pixel as last computed by the priority stage
                       // (valid while nothing inside the pixel's patch has changed since)
};
