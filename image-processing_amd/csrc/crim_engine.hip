// Host side of libcriminisi_hip.so: device memory, upload/download, hipGraph-driven fill loop
// and the C ABI declared in include/criminisi_hip.h.  Replaces the Python driver loop of
// Criminisi_Inpainter.inpaint() (/root/reference/src/criminisi_inpainter.py:40-88).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <math.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "crim_kernels.h"

// diagnostic stamp buffer: 64 named slots (plus per-block records in -DCRIM_STAMPS builds)
#ifdef CRIM_STAMPS
static constexpr size_t kDbgSlots = 64 + 8192;
#else
static constexpr size_t kDbgSlots = 64;
#endif

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(CRIM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// ------------------------------------------------------------------------------------------
// BGR -> Lab tables (OpenCV's integer RGB2Lab_b path, restated; the reference reaches it
// through cv2.cvtColor(..., COLOR_BGR2LAB), criminisi_inpainter.py:225)
// ------------------------------------------------------------------------------------------

namespace {

struct LabTables {
    uint16_t gamma[CRIM_GAMMA_N];
    uint16_t cbrt_[CRIM_CBRT_N];
    int32_t coef[9];
};

// cvRound: nearest, ties to even
int round_half_even(double v) { return (int)nearbyint(v); }

// OpenCV's cubeRoot(): exponent split, then a quartic rational polynomial in double, rounded to
// float.  Restated because the table entries are cvRound(32768 * this), not a true cbrt.
float opencv_cbrt(float value) {
    if (value == 0.0f) return 0.0f;
    uint32_t bits;
    memcpy(&bits, &value, 4);
    const uint32_t ix = bits & 0x7fffffffu;
    int ex = (int)(ix >> 23) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3;
    const uint32_t frbits = (ix & ((1u << 23) - 1u)) | ((uint32_t)(shx + 127) << 23);
    float frf;
    memcpy(&frf, &frbits, 4);
    const double fr = frf;
    const double num = ((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr +
                          119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr +
                        0.1636161226585754240958355063);
    const double den = ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr +
                          168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr +
                        1.0);
    float y = (float)(num / den);
    uint32_t yb;
    memcpy(&yb, &y, 4);
    yb += (uint32_t)(ex << 23);
    memcpy(&y, &yb, 4);
    return y;
}

void build_lab_tables(LabTables& t) {
    const float f255 = 255.0f;
    const float int_scale = 2040.0f;  // 255 * 2^3
    const double one_plus_shift = 1.0 + 11.0 / 200.0;
    for (int i = 0; i < 256; ++i) {
        const float x = (float)i / f255;
        const double xd = x;
        double g;
        if (xd <= 809.0 / 20000.0) g = xd / (323.0 / 25.0);
        else g = pow((xd + 11.0 / 200.0) / one_plus_shift, 12.0 / 5.0);
        const float gf = (float)g;
        t.gamma[i] = (uint16_t)round_half_even((double)(int_scale * gf));
    }
    const float lthresh = 216.0f / 24389.0f;
    const float lscale = 841.0f / 108.0f;
    const float lbias = 16.0f / 116.0f;
    const float cb_scale = 1.0f / (f255 * 8.0f);
    for (int i = 0; i < 3072; ++i) {
        const float x = cb_scale * (float)i;
        const float y = x < lthresh ? fmaf(x, lscale, lbias) : opencv_cbrt(x);
        t.cbrt_[i] = (uint16_t)round_half_even((double)(32768.0f * y));
    }
    static const double m[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                0.072169, 0.019334, 0.119193, 0.950227};
    static const double wp[3] = {0.950456, 1.0, 1.088754};
    for (int i = 0; i < 3; ++i) {
        t.coef[3 * i + 2] = round_half_even(4096.0 * m[3 * i + 0] / wp[i]);  // multiplies R
        t.coef[3 * i + 1] = round_half_even(4096.0 * m[3 * i + 1] / wp[i]);  // multiplies G
        t.coef[3 * i + 0] = round_half_even(4096.0 * m[3 * i + 2] / wp[i]);  // multiplies B
    }
}

const LabTables& lab_tables() {
    static LabTables t;
    static std::once_flag once;
    std::call_once(once, [] { build_lab_tables(t); });
    return t;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

// ------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------

struct GraphSet {
    hipGraphExec_t exec = nullptr;
    int iters = 0;
};

// A lane is a contiguous sub-batch that advances on its own stream: its latency-bound stages
// (priority, select, update) overlap the VALU-bound search of the other lanes.
struct Lane {
    hipStream_t stream = nullptr;
    Params prm;                     // engine parameters with b0 / B narrowed to this lane
    std::vector<GraphSet> graphs;
    long done = 0;                  // iterations launched in the current crim_run call
    long last = 0;                  // iterations launched in the previous round of that call
};

struct crim_engine {
    int device = 0;
    unsigned flags = 0;
    int patch_size = 0;
    Params prm;
    hipStream_t stream = nullptr;   // uploads, downloads, upload-time kernels
    std::vector<Lane> lanes;
    uint16_t* d_gamma = nullptr;
    uint16_t* d_cbrt = nullptr;
    ImgState* h_state = nullptr;    // pinned mirror of prm.st
    std::vector<bool> uploaded;
    std::vector<int> prev_remaining;   // per image: masked pixels left at the previous round of crim_run
    bool prepared = false;          // scan window, trace buffer and graphs match the uploads
    // profile accumulators
    std::vector<hipEvent_t> ev;     // pairs
    int ev_used = 0;
    std::vector<double> ev_evals;
    double prof_ms = 0.0, prof_evals = 0.0;
    long prof_launches = 0;
    void* d_peek = nullptr;         // scratch for crim_peek_fields
    // split search (one image, window-origin rows divided into parts)
    int split_parts = 0;            // 0: off
    int split_rank = -1;            // >= 0: this process scores part `rank` and the ranks exchange their bests
    ncclComm_t comm = nullptr;      // exchange by RCCL all-gather ...
    struct HostExchange* hx = nullptr;   // ... or through host memory shared by the ranks' processes
    std::vector<GraphSet> split_graphs;  // captured split iterations (64 / 8 / 1)
    bool split_capture_failed = false;   // stream capture of the split iteration was refused: the loop runs eagerly
    long split_exchanges = 0;            // records this process has exchanged through host memory
};

// Exchange of the ranks' shares of the search slots through a file-backed mapping shared by their
// processes (ranks on one host; a transport for tests and for placements RCCL refuses, e.g. two processes
// on ONE GPU).  Per rank and iteration parity one cell: a 64-byte header with the sequence number, then
// the share.  A rank publishes iteration k in cell k % 2 and cannot reach k + 2 before every rank has
// published k + 1, i.e. has read all cells of k: two cells per rank are enough.
struct HostExchange {
    int fd = -1;
    unsigned char* map = nullptr;
    size_t bytes = 0;
    size_t share = 0, cell_bytes = 0;   // bytes of one rank's share of the slots; of one cell
    int nranks = 0, rank = 0;
    unsigned long long seq = 0;         // iterations exchanged so far
    unsigned char* pinned = nullptr;    // page-locked staging: every rank's share, in rank order
    unsigned char* cell(int r, unsigned long long k) { return map + ((size_t)r * 2 + (k & 1)) * cell_bytes; }
};

// RCCL is loaded on demand so that the library has no link-time dependency on it.
namespace {
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;
int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return CRIM_OK;
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(CRIM_ERR_STATE, "split search: cannot load librccl.so: %s", dlerror());
    g_rccl.get_unique_id = (decltype(&ncclGetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (decltype(&ncclCommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.all_gather = (decltype(&ncclAllGather))dlsym(lib, "ncclAllGather");
    g_rccl.comm_destroy = (decltype(&ncclCommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.error_string = (decltype(&ncclGetErrorString))dlsym(lib, "ncclGetErrorString");
    g_rccl.comm_count = (decltype(&ncclCommCount))dlsym(lib, "ncclCommCount");
    g_rccl.comm_user_rank = (decltype(&ncclCommUserRank))dlsym(lib, "ncclCommUserRank");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_gather || !g_rccl.comm_destroy)
        return fail(CRIM_ERR_STATE, "split search: librccl.so lacks a required symbol");
    g_rccl.lib = lib;
    return CRIM_OK;
}
}  // namespace

static void free_graphs(crim_engine* h) {
    for (auto& ln : h->lanes) {
        for (auto& g : ln.graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        ln.graphs.clear();
    }
    for (auto& g : h->split_graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->split_graphs.clear();
}

static void close_host_exchange(crim_engine* h) {
    if (!h->hx) return;
    if (h->hx->pinned) (void)hipHostFree(h->hx->pinned);
    if (h->hx->map) munmap(h->hx->map, h->hx->bytes);
    if (h->hx->fd >= 0) close(h->hx->fd);
    delete h->hx;
    h->hx = nullptr;
}

// Search over block rows [sy0, sy0 + rows) of the window-origin grid (all rows when rows < 0).
// Does the search launch also do the select stage's work?  (only the pipelined matrix-core kernel can)
// The target selection as a launch of its own: one wave per image where the patch has at most 128 elements.
static void launch_select(const Params& p, hipStream_t s) {
    constexpr int NEL256 = (CRIM_PMAX * CRIM_PMAX + 255) / 256;
    if (p.P * p.P <= 128 && !getenv("CRIM_SELECT_256")) hipLaunchKernelGGL((k_select<64, 2>), dim3(p.B), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((k_select<256, NEL256>), dim3(p.B), dim3(256), 0, s, p);
}

static bool search_selects(const Params& p) {
    return p.selsearch && !p.generic && !p.dot4 && p.pipe_t > 0 && p.P >= 3 && p.P <= 15 && (p.P & 1);
}

// Every instance of the matrix-core search kernels, looked up by (patch edge, colour planes, selects):
// one table per patch edge, so that a launch site cannot name a wrong <P, CHROMA, SELECT> triple.
typedef void (*SearchKernel)(Params);
template <int P>
struct SearchInstances {
    static constexpr SearchKernel pipe[2][2] = {{k_search_pipe<P, false, false>, k_search_pipe<P, false, true>},
                                               {k_search_pipe<P, true, false>, k_search_pipe<P, true, true>}};
    static constexpr SearchKernel mfma[2] = {k_search_mfma<P, false>, k_search_mfma<P, true>};
};
template <template <int> class F, typename R, typename... A>
static R by_patch_edge(int P, R none, A... a) {
    switch (P) {
        case 3: return F<3>::get(a...);
        case 5: return F<5>::get(a...);
        case 7: return F<7>::get(a...);
        case 9: return F<9>::get(a...);
        case 11: return F<11>::get(a...);
        case 13: return F<13>::get(a...);
        case 15: return F<15>::get(a...);
        default: return none;
    }
}
template <int P> struct PipeGet { static SearchKernel get(bool chroma, bool sel) { return SearchInstances<P>::pipe[chroma][sel]; } };
template <int P> struct MfmaGet { static SearchKernel get(bool chroma) { return SearchInstances<P>::mfma[chroma]; } };
static SearchKernel pipe_kernel(int P, bool chroma, bool sel) { return by_patch_edge<PipeGet, SearchKernel>(P, nullptr, chroma, sel); }
static SearchKernel mfma_kernel(int P, bool chroma) { return by_patch_edge<MfmaGet, SearchKernel>(P, nullptr, chroma); }

static void launch_search(const Params& pin, hipStream_t s, int sy0 = 0, int rows = -1, bool sel = false) {
    Params p = pin;
    p.sy0 = sy0;
    if (rows < 0) rows = p.sny;
    if (rows == 0) return;
    dim3 grid(p.snx, rows, p.B), block(256);
    if (p.generic) {
        hipLaunchKernelGGL(k_search_generic, grid, block, 0, s, p);
        return;
    }
    if (!p.dot4) {
        // SB_H == MS_ROWS: the block rows of both specialised kernels coincide
        p.mgx = (std::max(p.W - p.half, 1) + MS_W - 1) / MS_W;
        p.mrows = rows;
        const int pipe_t = p.pipe_t;
        if (pipe_t > 0) {
            // pipelined kernel: up to pipe_t rows of tiles per block, fewer when the launch would
            // otherwise have fewer blocks than the GPU holds at once (256 CUs x 4)
            int T = std::min(pipe_t, rows);
            const long min_blocks = p.pipe_min_blocks;
            while (T > 1 && (long)p.mgx * ((rows + T - 1) / T) * p.B < min_blocks) T = (T + 1) / 2;
            if (p.pipe_force > 0) T = std::min(p.pipe_force, rows);
            p.mtiles = T;
            p.mchunks = (rows + T - 1) / T;
            dim3 pgrid((unsigned)p.mgx * p.mchunks * p.B);
            if (const SearchKernel k = pipe_kernel(p.P, p.chroma != 0, sel)) {
                hipLaunchKernelGGL(k, pgrid, block, 0, s, p);
                return;
            }
        }
        dim3 mgrid((unsigned)p.mgx * rows * p.B);
        if (const SearchKernel k = mfma_kernel(p.P, p.chroma != 0)) {
            hipLaunchKernelGGL(k, mgrid, block, 0, s, p);
            return;
        }
    }
    switch (p.P) {
        case 3: hipLaunchKernelGGL(k_search<3>, grid, block, 0, s, p); break;
        case 5: hipLaunchKernelGGL(k_search<5>, grid, block, 0, s, p); break;
        case 7: hipLaunchKernelGGL(k_search<7>, grid, block, 0, s, p); break;
        case 9: hipLaunchKernelGGL(k_search<9>, grid, block, 0, s, p); break;
        case 11: hipLaunchKernelGGL(k_search<11>, grid, block, 0, s, p); break;
        case 13: hipLaunchKernelGGL(k_search<13>, grid, block, 0, s, p); break;
        case 15: hipLaunchKernelGGL(k_search<15>, grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL(k_search_generic, grid, block, 0, s, p); break;
    }
}

static bool has_fast_search(int P) { return P >= 3 && P <= 15 && (P & 1); }

// Tile-list rebuild: stale entries only cost a skipped lookup, so it runs rarely.
static void launch_compact(const Params& p, hipStream_t s) {
    hipLaunchKernelGGL(k_tiles, dim3(p.B), dim3(1024), 0, s, p);
}

static size_t priority_lds_bytes(const Params& p) {
    const size_t rs = CRIM_TILE + 2 * p.half;
    // two 8-byte planes, front list (bytes), mask, per-lane offset table (4 lanes x P*P/8 pairs of 16-bit offsets
    // on the fast path, patch edges up to 15), alignment slack.  At patch edge 9 this is just under 10 KB.
    const size_t table = 4 * (size_t)(p.P <= 15 ? (p.P * p.P) / 8 + 1 : 1) * 4;
    return rs * rs * 16 + CRIM_TILE * CRIM_TILE + (CRIM_TILE + 2) * (CRIM_TILE + 2) + table + 8;
}

static void launch_update(const Params& p, hipStream_t s) {
    if (p.P <= 9) hipLaunchKernelGGL(k_update<9>, dim3(p.B), dim3(256), 0, s, p);
    else if (p.P <= 15) hipLaunchKernelGGL(k_update<15>, dim3(p.B), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_update<CRIM_PMAX>, dim3(p.B), dim3(256), 0, s, p);
}

static void launch_priority(const Params& p, hipStream_t s) {
    const dim3 grid((unsigned)p.pgrid * p.B), block(64);
    const size_t lds = priority_lds_bytes(p);
    switch (p.P) {
        case 3: hipLaunchKernelGGL(k_priority<3>, grid, block, lds, s, p); break;
        case 5: hipLaunchKernelGGL(k_priority<5>, grid, block, lds, s, p); break;
        case 7: hipLaunchKernelGGL(k_priority<7>, grid, block, lds, s, p); break;
        case 9: hipLaunchKernelGGL(k_priority<9>, grid, block, lds, s, p); break;
        case 11: hipLaunchKernelGGL(k_priority<11>, grid, block, lds, s, p); break;
        case 13: hipLaunchKernelGGL(k_priority<13>, grid, block, lds, s, p); break;
        case 15: hipLaunchKernelGGL(k_priority<15>, grid, block, lds, s, p); break;
        default: hipLaunchKernelGGL(k_priority<0>, grid, block, lds, s, p); break;
    }
}

// The launch that follows the search in the two-launch iteration (k_step: update + the next iteration's
// priority stage); only patch edges with a pipelined search instance reach it.
static void launch_step(const Params& p, hipStream_t s) {
    const dim3 grid((unsigned)p.kblocks * p.B), block(256);
    const size_t lds = (size_t)4 * p.plds;
    switch (p.P) {
        case 3: hipLaunchKernelGGL(k_step<3>, grid, block, lds, s, p); break;
        case 5: hipLaunchKernelGGL(k_step<5>, grid, block, lds, s, p); break;
        case 7: hipLaunchKernelGGL(k_step<7>, grid, block, lds, s, p); break;
        case 9: hipLaunchKernelGGL(k_step<9>, grid, block, lds, s, p); break;
        case 11: hipLaunchKernelGGL(k_step<11>, grid, block, lds, s, p); break;
        case 13: hipLaunchKernelGGL(k_step<13>, grid, block, lds, s, p); break;
        default: hipLaunchKernelGGL(k_step<15>, grid, block, lds, s, p); break;
    }
}

// Iterations between two ordered rebuilds of the tile list (stale entries only cost a skipped
// lookup; a rebuild walks every tile of the image, so large images rebuild less often).
static int compact_period(const Params& p) { return p.tnx * p.tny > 1024 ? 64 : 16; }

// One iteration of criminisi_inpainter.py:51-83 as four dependent launches.  ev0/ev1, when
// given, bracket the search launch.
static void launch_iteration(const Params& p, hipStream_t s, hipEvent_t ev0 = nullptr,
                             hipEvent_t ev1 = nullptr) {
    launch_priority(p, s);
    const bool sel = search_selects(p);
    if (!sel) launch_select(p, s);
    if (ev0) (void)hipEventRecord(ev0, s);
    launch_search(p, s, 0, -1, sel);
    if (ev1) (void)hipEventRecord(ev1, s);
    launch_update(p, s);
}

// The two-launch iteration (p.fused): the search launch, which also selects the target, then k_step.
// The priority stage of the FIRST iteration of a sequence is launch_priority, issued once by the caller
// (it is idempotent: repeating it for an iteration whose priorities k_step has already worked out
// rewrites the same values).
static void launch_iteration_fused(const Params& p, hipStream_t s) {
    launch_search(p, s, 0, -1, true);
    launch_step(p, s);
}

static int build_graph(Lane& ln, int iters, GraphSet* out) {
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(ln.stream, hipStreamCaptureModeThreadLocal));
    const int period = compact_period(ln.prm);
    for (int i = 0; i < iters; ++i) {
        if (ln.prm.fused) launch_iteration_fused(ln.prm, ln.stream);
        else launch_iteration(ln.prm, ln.stream);
        if ((i + 1) % period == 0) launch_compact(ln.prm, ln.stream);
    }
    HIP_TRY(hipStreamEndCapture(ln.stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(CRIM_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    out->exec = exec;
    out->iters = iters;
    return CRIM_OK;
}

// Everything of Params that follows from the shape alone (no device memory).
static void init_geometry(Params& p, int batch, int H, int W, int patch_size, unsigned flags) {
    memset(&p, 0, sizeof(p));
    p.B = batch; p.H = H; p.W = W;
    p.half = (patch_size - 1) / 2;
    p.P = 2 * p.half + 1;
    p.psq = patch_size * patch_size;
    p.ND = (p.P + 6) / 4;
    p.Wp = round_up(W + 16, 16);
    p.Wi = round_up(W, 4) + 4;
    p.Wb = 4 * ((W + MS_W - 1) / MS_W);
    p.dmax = (int)std::ceil(std::sqrt((double)H * H + (double)W * W)) + 1;
    p.tnx = (W + CRIM_TILE - 1) / CRIM_TILE;
    p.tny = (H + CRIM_TILE - 1) / CRIM_TILE;
    p.tcap = 2 * p.tnx * p.tny;
    // the candidate grid is largest for the most clipped patch: (H - half) x (W - half)
    p.snx = (std::max(W - p.half, 1) + SB_W - 1) / SB_W;
    p.sny = (std::max(H - p.half, 1) + SB_H - 1) / SB_H;
    p.srows = p.sny + 64;   // a split search exchanges equal shares of whole block rows: up to 64 shares
    p.generic = ((flags & CRIM_FLAG_GENERIC_SEARCH) || !has_fast_search(p.P)) ? 1 : 0;
    p.dot4 = (flags & CRIM_FLAG_DOT4_SEARCH) ? 1 : 0;
    // a candidate grid is at most W - half origins wide; the matrix-core kernels tile it 256 wide
    p.scols = (p.generic || p.dot4) ? p.snx : (std::max(W - p.half, 1) + MS_W - 1) / MS_W;
    {
        // tuning knobs of the pipelined grey search, read when the engine is created
        const char* e = getenv("CRIM_PIPE_T");
        p.pipe_t = e ? atoi(e) : 8;
        e = getenv("CRIM_PIPE_MIN_BLOCKS");
        p.pipe_min_blocks = e ? atoi(e) : 512;
        e = getenv("CRIM_PIPE_TILES");      // experiments: exactly this many rows of tiles per block
        p.pipe_force = e ? atoi(e) : 0;
    }
    // priority waves per image: a wave's fixed cost (state, set-up: about 3 us) is shared by the tiles it
    // walks, so large batches, which fill the GPU anyway, get few waves per image
    p.pgrid = std::min(std::min(2048, p.tnx * p.tny), std::max(8, 16384 / batch));
    if (const char* e = getenv("CRIM_PGRID")) {   // tuning knob: priority waves per image
        const int g = atoi(e);
        if (g > 0) p.pgrid = std::min(g, p.tnx * p.tny);
    }
    p.b0 = 0;
    p.trace_cap = 0;
    // Small batches wait for their launch boundaries and round trips: the search launch then also picks
    // the target and builds the template (one launch less per iteration); every search block reads the
    // priority stage's slots, so there are fewer of them.
    p.selsearch = batch <= 2048 ? 1 : 0;   // measured: +9 % at 100 images, +4 % at 1024, -3 % at 8192
    if (const char* e = getenv("CRIM_SELECT_IN_SEARCH")) p.selsearch = atoi(e) ? 1 : 0;
    if (flags & (CRIM_FLAG_PROFILE | CRIM_FLAG_PROFILE_SOLO)) p.selsearch = 0;   // the bracketed launch is the search alone
    if (p.selsearch && !getenv("CRIM_PGRID")) p.pgrid = std::min(p.pgrid, 1024);   // every search block reads all of them
    p.plds = (int)((priority_lds_bytes(p) + 15) & ~(size_t)15);
    // Two launches per iteration where the search launch selects: the update and the next iteration's
    // priority stage share one launch (k_step).  Its priority slots: one per near tile, then one per
    // far wave (a multiple of 4: four waves per block), at most 1024 in all (what a search block reads
    // in one batch of loads).
    p.fused = (search_selects(p) && !(flags & (CRIM_FLAG_EAGER | CRIM_FLAG_PROFILE | CRIM_FLAG_PROFILE_SOLO))) ? 1 : 0;
    // images a launch covers: a batch of 32 or more advances as two lanes (crim_create_batch)
    const int lane_batch = (batch >= 32 && !(flags & CRIM_FLAG_ONE_LANE)) ? (batch + 1) / 2 : batch;
    // It pays where iterations wait for launches and round trips, not where the launches fill the GPU many
    // times over (measured: 100 x 256^2 -20 % per iteration, 800 x 512^2 +5 %): up to 4096 search tiles per launch.
    const long search_tiles = (long)lane_batch * ((std::max(W - p.half, 1) + MS_W - 1) / MS_W) * p.sny;
    if (search_tiles > 4096) p.fused = 0;
    if (const char* e = getenv("CRIM_FUSED")) p.fused = (search_selects(p) && atoi(e)) ? 1 : 0;
    if (p.fused) {
        // far waves per image: as many as the tile list may need (every wave then has at most one tile), at
        // least 28, at most what the selecting search reads in one batch of loads; for batches about 1024
        // blocks per launch (most near blocks leave at once).  Measured on 50 images per launch: 20 far waves
        // 39.5 us per lock-step iteration, 28 to 60 waves 39.0-39.1 us, 12 waves 43 us.
        int far_blocks = 1024 / lane_batch - 1 - CRIM_NEAR_TILES;
        far_blocks = std::max(7, std::min(far_blocks, (1024 - CRIM_NEAR_TILES - 3) / 4));
        far_blocks = std::min(far_blocks, (p.tnx * p.tny + 3) / 4);
        p.nfar = 4 * far_blocks;
        if (getenv("CRIM_PGRID")) p.nfar = std::max(4, std::min(p.pgrid, 1024 - CRIM_NEAR_TILES - 3) / 4 * 4);
        p.pgrid = CRIM_NEAR_TILES + p.nfar;
        p.kblocks = 1 + CRIM_NEAR_TILES + p.nfar / 4;
    }
}

extern "C" int crim_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* crim_last_error(void) { return g_err; }

extern "C" int crim_lab_tables(uint16_t* gamma256, uint16_t* cbrt3072, int32_t* coeffs9) {
    const LabTables& t = lab_tables();
    if (gamma256) memcpy(gamma256, t.gamma, sizeof(t.gamma));
    if (cbrt3072) memcpy(cbrt3072, t.cbrt_, sizeof(t.cbrt_));
    if (coeffs9) memcpy(coeffs9, t.coef, sizeof(t.coef));
    return CRIM_OK;
}

extern "C" void crim_destroy(crim_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& ln : h->lanes)
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
    free_graphs(h);
    if (h->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(h->comm);
    close_host_exchange(h);
    for (auto& ln : h->lanes)
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    for (auto e : h->ev)
        if (e) (void)hipEventDestroy(e);
    Params& p = h->prm;
    void* bufs[] = {p.bgr, p.planes, p.mask, p.grey, p.grad, p.conf, p.dterm, p.invalid, p.invbits, p.tfront, p.tlist, p.st,
                    p.desc, p.plist, p.slist, p.trace, p.stime, p.dbg, h->d_gamma, h->d_cbrt, h->d_peek};
    for (void* q : bufs)
        if (q) (void)hipFree(q);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int crim_create_batch(int batch, int H, int W, int patch_size, int device,
                                 unsigned flags, crim_t** out) {
    if (!out) return fail(CRIM_ERR_ARG, "crim_create: out is null");
    *out = nullptr;
    if (batch < 1 || H < 1 || W < 1) return fail(CRIM_ERR_ARG, "crim_create: bad shape %d x %d x %d", batch, H, W);
    if (patch_size < 1 || patch_size > CRIM_MAX_PATCH)
        return fail(CRIM_ERR_ARG, "crim_create: patch_size %d outside 1..%d", patch_size, CRIM_MAX_PATCH);
    if ((double)H * W > 2.0e9 / 4) return fail(CRIM_ERR_ARG, "crim_create: image too large");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CRIM_ERR_ARG, "crim_create: device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    crim_engine* h = new (std::nothrow) crim_engine();
    if (!h) return fail(CRIM_ERR_NOMEM, "crim_create: out of host memory");
    h->device = device;
    h->flags = flags;
    if (flags & CRIM_FLAG_PROFILE_SOLO) h->flags |= CRIM_FLAG_PROFILE;
    if (h->flags & CRIM_FLAG_PROFILE) h->flags |= CRIM_FLAG_EAGER;
    h->patch_size = patch_size;
    Params& p = h->prm;
    init_geometry(p, batch, H, W, patch_size, flags);
    const size_t HW = (size_t)H * W, Bz = (size_t)batch;
    const LabTables& t = lab_tables();
    memcpy(p.coef, t.coef, sizeof(p.coef));
    hipError_t e = hipSuccess;
    auto alloc = [&](void** ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(ptr, bytes);
    };
    alloc((void**)&p.bgr, Bz * HW * 3);
    // 64 spare rows: the pipelined search reads whole row tiles, also below the last plane's last row
    alloc((void**)&p.planes, Bz * CRIM_NPL * H * p.Wp + (size_t)64 * p.Wp);
    alloc((void**)&p.mask, Bz * HW);
    alloc((void**)&p.grey, Bz * HW);
    alloc((void**)&p.grad, Bz * HW * sizeof(int));
    alloc((void**)&p.conf, Bz * 2 * HW * sizeof(double));
    alloc((void**)&p.dterm, Bz * HW * sizeof(double));
    alloc((void**)&p.invalid, Bz * H * p.Wi);
    alloc((void**)&p.invbits, Bz * H * p.Wb * sizeof(unsigned long long));
    alloc((void**)&p.tfront, Bz * p.tnx * p.tny * sizeof(int));
    alloc((void**)&p.tlist, Bz * 2 * p.tcap * sizeof(int));
    alloc((void**)&p.st, Bz * sizeof(ImgState));
    alloc((void**)&p.desc, Bz * sizeof(Desc));
    alloc((void**)&p.plist, Bz * p.pgrid * sizeof(PEntry));
    alloc((void**)&p.slist, Bz * p.snx * p.srows * sizeof(SEntry));
    alloc((void**)&p.dbg, kDbgSlots * sizeof(long long));
    alloc((void**)&h->d_gamma, sizeof(t.gamma));
    alloc((void**)&h->d_cbrt, sizeof(t.cbrt_));
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_state, Bz * sizeof(ImgState), hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpy(h->d_gamma, t.gamma, sizeof(t.gamma), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_cbrt, t.cbrt_, sizeof(t.cbrt_), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(p.planes, 0, Bz * CRIM_NPL * H * p.Wp + (size_t)64 * p.Wp);
    if (e == hipSuccess) e = hipMemset(p.invalid, 1, Bz * H * p.Wi);
    if (e == hipSuccess) e = hipMemset(p.invbits, 0xff, Bz * H * p.Wb * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(p.st, 0, Bz * sizeof(ImgState));
    if (e == hipSuccess) e = hipMemset(p.tlist, 0, Bz * 2 * p.tcap * sizeof(int));
    if (e == hipSuccess) e = hipMemset(p.dbg, 0, kDbgSlots * sizeof(long long));
    if (e == hipSuccess) e = hipMemset(p.desc, 0, Bz * sizeof(Desc));
    // (memsets of device memory need not have finished when they return, and the engine's streams do not wait
    // for the stream they ran on: an upload right after this call must not overtake them)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        const int code = fail(e == hipErrorOutOfMemory ? CRIM_ERR_NOMEM : CRIM_ERR_HIP,
                              "crim_create: %s", hipGetErrorString(e));
        crim_destroy(h);
        return code;
    }
    p.gamma_tab = h->d_gamma;
    p.cbrt_tab = h->d_cbrt;
    h->uploaded.assign(batch, false);
    {
        // Lanes: sub-batches on their own streams (one lane when the batch is small).  Measured on
        // 256 images of 256x256: 2 lanes +9 % images/s over 1, 4 lanes -17 % (the streams share
        // two hardware queues and every lane adds its own latency-bound launches).
        int nl = batch >= 32 ? 2 : 1;
        if (const char* env = getenv("CRIM_LANES")) nl = atoi(env);
        if (flags & CRIM_FLAG_ONE_LANE) nl = 1;
        nl = std::max(1, std::min(nl, std::min(batch, 16)));
        h->lanes.resize(nl);
        for (int i = 0; i < nl; ++i) {
            Lane& ln = h->lanes[i];
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking);
            ln.prm = p;
            ln.prm.b0 = (int)((long)batch * i / nl);
            ln.prm.B = (int)((long)batch * (i + 1) / nl) - ln.prm.b0;
        }
    }
    if (h->flags & CRIM_FLAG_PROFILE) {
        h->ev.resize(2 * 64);
        h->ev_evals.resize(64);
        for (auto& ev : h->ev) ev = nullptr;
        for (auto& ev : h->ev)
            if (e == hipSuccess) e = hipEventCreate(&ev);
    }
    if (e != hipSuccess) {
        const int code = fail(CRIM_ERR_HIP, "crim_create: %s", hipGetErrorString(e));
        crim_destroy(h);
        return code;
    }
    *out = h;
    return CRIM_OK;
}

extern "C" int crim_create(int H, int W, int patch_size, int device, unsigned flags, crim_t** out) {
    return crim_create_batch(1, H, W, patch_size, device, flags, out);
}

// Derives every cached field of images [first, first+count) from prm.bgr / prm.mask.
// (src_bgr / src_mask: dense device images the pixel kernel reads instead of prm.bgr / prm.mask, which it then fills)
static void launch_init(const Params& p, hipStream_t s, int first, int count, const uint8_t* src_bgr = nullptr,
                        const uint8_t* src_mask = nullptr) {
    hipLaunchKernelGGL(k_init_state, dim3((count + 63) / 64), dim3(64), 0, s, p, first, count);
    dim3 grid((p.W + IF_W - 1) / IF_W, p.tny, count), block(256);   // k_init_fields: strips of 16 rows x IF_W columns
    const size_t HW = (size_t)p.H * p.W;
    hipLaunchKernelGGL(k_init_pixels, dim3((unsigned)((HW + 1023) / 1024), count), block, 0, s, p, first, src_bgr, src_mask);
    hipLaunchKernelGGL(k_init_fields, grid, block, 0, s, p, first);
}

static int init_images(crim_engine* h, int first, int count, const uint8_t* src_bgr = nullptr, const uint8_t* src_mask = nullptr) {
    const Params& p = h->prm;
    launch_init(p, h->stream, first, count, src_bgr, src_mask);
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < count; ++i) h->uploaded[first + i] = true;
    h->prepared = false;
    h->prof_ms = 0.0; h->prof_evals = 0.0; h->prof_launches = 0;
    return CRIM_OK;
}

extern "C" int crim_upload_batch(crim_t* h, int first, int count, const uint8_t* bgr,
                                 ptrdiff_t image_stride, ptrdiff_t row_stride, const uint8_t* mask,
                                 ptrdiff_t mask_image_stride, ptrdiff_t mask_row_stride) {
    if (!h || !bgr || !mask) return fail(CRIM_ERR_ARG, "crim_upload: null argument");
    const Params& p = h->prm;
    if (first < 0 || count < 1 || first + count > p.B)
        return fail(CRIM_ERR_ARG, "crim_upload: images [%d,%d) outside batch of %d", first, first + count, p.B);
    if (row_stride < (ptrdiff_t)p.W * 3 || mask_row_stride < (ptrdiff_t)p.W)
        return fail(CRIM_ERR_ARG, "crim_upload: row stride smaller than a row");
    HIP_TRY(hipSetDevice(h->device));
    const size_t HW = (size_t)p.H * p.W;
    // dense inputs travel as one copy each (thousands of per-image copies cost more host time than
    // the transfer itself)
    const bool dense_img = row_stride == (ptrdiff_t)p.W * 3 && (count == 1 || image_stride == (ptrdiff_t)(HW * 3));
    const bool dense_msk = mask_row_stride == (ptrdiff_t)p.W && (count == 1 || mask_image_stride == (ptrdiff_t)HW);
    if (dense_img)
        HIP_TRY(hipMemcpyAsync(p.bgr + (size_t)first * HW * 3, bgr, (size_t)count * HW * 3, hipMemcpyHostToDevice, h->stream));
    if (dense_msk)
        HIP_TRY(hipMemcpyAsync(p.mask + (size_t)first * HW, mask, (size_t)count * HW, hipMemcpyHostToDevice, h->stream));
    for (int i = 0; i < count; ++i) {
        if (!dense_img)
            HIP_TRY(hipMemcpy2DAsync(p.bgr + (size_t)(first + i) * HW * 3, (size_t)p.W * 3,
                                     bgr + (ptrdiff_t)i * image_stride, (size_t)row_stride,
                                     (size_t)p.W * 3, p.H, hipMemcpyHostToDevice, h->stream));
        if (!dense_msk)
            HIP_TRY(hipMemcpy2DAsync(p.mask + (size_t)(first + i) * HW, (size_t)p.W,
                                     mask + (ptrdiff_t)i * mask_image_stride, (size_t)mask_row_stride,
                                     (size_t)p.W, p.H, hipMemcpyHostToDevice, h->stream));
    }
    const int rc = init_images(h, first, count);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));   // the host buffers may be reused on return
    return CRIM_OK;
}

extern "C" int crim_upload_automask(crim_t* h, int first, int count, const uint8_t* bgr,
                                    ptrdiff_t image_stride, ptrdiff_t row_stride, int threshold,
                                    int border_width, int dilation_size) {
    if (!h || !bgr) return fail(CRIM_ERR_ARG, "crim_upload_automask: null argument");
    const Params& p = h->prm;
    if (first < 0 || count < 1 || first + count > p.B)
        return fail(CRIM_ERR_ARG, "crim_upload_automask: images [%d,%d) outside batch of %d", first, first + count, p.B);
    if (row_stride < (ptrdiff_t)p.W * 3) return fail(CRIM_ERR_ARG, "crim_upload_automask: row stride smaller than a row");
    if (threshold < 0 || threshold > 255 || border_width < 0 || dilation_size < 1 || dilation_size > 64)
        return fail(CRIM_ERR_ARG, "crim_upload_automask: threshold %d, border %d or dilation %d out of range",
                    threshold, border_width, dilation_size);
    HIP_TRY(hipSetDevice(h->device));
    const size_t HW = (size_t)p.H * p.W;
    if (row_stride == (ptrdiff_t)p.W * 3 && (count == 1 || image_stride == (ptrdiff_t)(HW * 3)))
        HIP_TRY(hipMemcpyAsync(p.bgr + (size_t)first * HW * 3, bgr, (size_t)count * HW * 3, hipMemcpyHostToDevice, h->stream));
    else
        for (int i = 0; i < count; ++i)
            HIP_TRY(hipMemcpy2DAsync(p.bgr + (size_t)(first + i) * HW * 3, (size_t)p.W * 3,
                                     bgr + (ptrdiff_t)i * image_stride, (size_t)row_stride,
                                     (size_t)p.W * 3, p.H, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_build_mask, dim3(p.tnx, p.tny, count), dim3(256), 0, h->stream, p, first,
                       threshold, border_width, dilation_size);
    const int rc = init_images(h, first, count);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CRIM_OK;
}

// Page-locked host memory: uploads from it and downloads into it are asynchronous DMA transfers
// that overlap with other engines' kernels.
extern "C" int crim_host_alloc(size_t bytes, void** out) {
    if (!out || !bytes) return fail(CRIM_ERR_ARG, "crim_host_alloc: null argument");
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(e == hipErrorOutOfMemory ? CRIM_ERR_NOMEM : CRIM_ERR_HIP, "crim_host_alloc(%zu): %s", bytes,
                    hipGetErrorString(e));
    }
    return CRIM_OK;
}

extern "C" int crim_host_free(void* ptr) {
    if (!ptr) return CRIM_OK;
    HIP_TRY(hipHostFree(ptr));
    return CRIM_OK;
}

extern "C" int crim_upload(crim_t* h, const uint8_t* bgr, ptrdiff_t row_stride, const uint8_t* mask,
                           ptrdiff_t mask_row_stride) {
    return crim_upload_batch(h, 0, 1, bgr, 0, row_stride, mask, 0, mask_row_stride);
}

extern "C" int crim_upload_device(crim_t* h, int first, int count, const void* d_bgr,
                                  const void* d_mask) {
    if (!h || !d_bgr || !d_mask) return fail(CRIM_ERR_ARG, "crim_upload_device: null argument");
    const Params& p = h->prm;
    if (first < 0 || count < 1 || first + count > p.B)
        return fail(CRIM_ERR_ARG, "crim_upload_device: images [%d,%d) outside batch of %d", first, first + count, p.B);
    HIP_TRY(hipSetDevice(h->device));
    const size_t HW = (size_t)p.H * p.W;
    // (no copy beforehand: the kernel that derives the planes reads the caller's buffers and fills the engine's;
    // the caller's buffers must stay untouched until the stream has passed the upload, as with the copy before)
    const uint8_t* sb = (const uint8_t*)d_bgr;
    const uint8_t* sm = (const uint8_t*)d_mask;
    if (sb == p.bgr + (size_t)first * HW * 3) sb = nullptr;      // already in place
    if (sm == p.mask + (size_t)first * HW) sm = nullptr;
    const bool aligned = ((uintptr_t)d_bgr & 3) == 0 && ((uintptr_t)d_mask & 3) == 0;   // the kernel reads dwords
    if ((sb == nullptr) != (sm == nullptr) || !aligned) {        // one of the two in place, or odd addresses: copy as before
        if (sb) HIP_TRY(hipMemcpyAsync(p.bgr + (size_t)first * HW * 3, sb, (size_t)count * HW * 3, hipMemcpyDeviceToDevice, h->stream));
        if (sm) HIP_TRY(hipMemcpyAsync(p.mask + (size_t)first * HW, sm, (size_t)count * HW, hipMemcpyDeviceToDevice, h->stream));
        sb = sm = nullptr;
    }
    return init_images(h, first, count, sb, sm);
}

// Sizes the trace buffer from what was uploaded, then (re)builds the iteration graphs.
static int prepare(crim_engine* h) {
    Params& p = h->prm;
    for (int i = 0; i < p.B; ++i)
        if (!h->uploaded[i]) return fail(CRIM_ERR_STATE, "crim_run: image %d was never uploaded", i);
    HIP_TRY(hipMemcpyAsync(h->h_state, p.st, (size_t)p.B * sizeof(ImgState), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    Params q = p;
    long need = 1;
    for (int i = 0; i < p.B; ++i) need = std::max(need, (long)h->h_state[i].remaining + 1);
    q.chroma = 0;
    for (int i = 0; i < p.B; ++i) q.chroma |= h->h_state[i].chroma ? 1 : 0;
    if (!(h->flags & CRIM_FLAG_NO_TRACE) && (need > p.trace_cap || !p.trace)) {
        if (p.trace) HIP_TRY(hipFree(p.trace));
        if (p.stime) HIP_TRY(hipFree(p.stime));
        q.trace = nullptr;
        p.trace = nullptr;
        q.stime = nullptr;
        p.stime = nullptr;
        q.trace_cap = 0;
        p.trace_cap = 0;
        const hipError_t te = hipMalloc((void**)&q.trace, (size_t)p.B * need * sizeof(crim_iter_rec));
        if (te != hipSuccess) {
            q.trace = nullptr;
            return fail(te == hipErrorOutOfMemory ? CRIM_ERR_NOMEM : CRIM_ERR_HIP,
                        "crim_run: per-iteration records (%ld per image, %d images): %s; "
                        "create the engine with CRIM_FLAG_NO_TRACE to run without them",
                        need, p.B, hipGetErrorString(te));
        }
        q.trace_cap = need;
        if (hipMalloc((void**)&q.stime, (size_t)p.B * need * 3 * sizeof(long long)) != hipSuccess) q.stime = nullptr;   // optional
    }
    const bool changed = memcmp(&q, &p, sizeof(Params)) != 0;
    p = q;
    bool have_graphs = true;
    for (auto& ln : h->lanes) have_graphs = have_graphs && !ln.graphs.empty();
    if (changed || !have_graphs) {
        free_graphs(h);
        for (auto& ln : h->lanes) {
            const int b0 = ln.prm.b0, nb = ln.prm.B;
            ln.prm = p;
            ln.prm.b0 = b0;
            ln.prm.B = nb;
            if (h->flags & CRIM_FLAG_EAGER) continue;
            static const int sizes[3] = {64, 8, 1};
            for (int s : sizes) {
                GraphSet g;
                const int rc = build_graph(ln, s, &g);
                if (rc) return rc;
                ln.graphs.push_back(g);
            }
        }
    }
    h->prepared = true;
    return CRIM_OK;
}

static int flush_profile(crim_engine* h) {
    for (int i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(h->ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
        h->prof_ms += ms;
        h->prof_evals += h->ev_evals[i];
        h->prof_launches += 1;
    }
    h->ev_used = 0;
    return CRIM_OK;
}

// The split of the search: block rows [y0, y1) of part `part` of `parts` -- equal shares of whole block
// rows (the last ones may be short or empty), so that the parts' slots are equal, contiguous pieces of the
// image's slot array and an all-gather can put them together in place.
static int split_rows_per_part(const Params& p, int parts) { return (p.sny + parts - 1) / parts; }

// One iteration of the fill loop with the search split into parts (one image): the search of this
// process's part(s); when the parts live in different processes, the exchange of their slots (RCCL
// all-gather on the stream, in place: afterwards every process holds every search block's best window,
// exactly as after an unsplit search); the update, which reduces the slots as it always does.
// Where the search launch can select the target itself it does (every part's launch picks the same
// target), and the update shares its launch with the next iteration's priority stage (k_step), exactly
// as in the unsplit loop; the sequence then starts with ONE priority launch issued by the caller.
// `first_half` / `second_half`: a host-staged exchange sits between the two.
static int launch_split_iteration(crim_engine* h, const Params& p, hipStream_t s, bool first_half, bool second_half) {
    const int parts = h->split_parts;
    const bool sel = search_selects(p);
    const int rpp = split_rows_per_part(p, parts);
    if (first_half) {
        if (!p.fused) launch_priority(p, s);
        if (!sel) launch_select(p, s);
        for (int part = 0; part < parts; ++part) {
            if (h->split_rank >= 0 && part != h->split_rank) continue;
            const int y0 = std::min(part * rpp, p.sny), y1 = std::min(y0 + rpp, p.sny);
            launch_search(p, s, y0, y1 - y0, sel);
        }
        if (h->split_rank >= 0 && h->comm) {
            const size_t share = (size_t)rpp * p.snx * sizeof(SEntry);
            const ncclResult_t r = g_rccl.all_gather((const char*)p.slist + (size_t)h->split_rank * share, p.slist, share,
                                                     ncclInt8, h->comm, s);
            if (r != ncclSuccess)
                return fail(CRIM_ERR_HIP, "ncclAllGather: %s", g_rccl.error_string ? g_rccl.error_string(r) : "error");
        }
    }
    if (second_half) {
        if (p.fused) launch_step(p, s);
        else launch_update(p, s);
    }
    return CRIM_OK;
}

// Publishes this rank's share of the current iteration's slots and collects every other rank's (host
// memory shared by the ranks' processes).  Gives up after 60 s without progress: a peer that died must
// not hang the others.
static int host_exchange(crim_engine* h, const Params& p, hipStream_t s) {
    HostExchange& x = *h->hx;
    unsigned char* d_slots = (unsigned char*)p.slist;
    HIP_TRY(hipMemcpyAsync(x.pinned + (size_t)x.rank * x.share, d_slots + (size_t)x.rank * x.share, x.share,
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const unsigned long long k = ++x.seq;
    unsigned char* mine = x.cell(x.rank, k);
    memcpy(mine + 64, x.pinned + (size_t)x.rank * x.share, x.share);
    __atomic_store_n((unsigned long long*)mine, k, __ATOMIC_RELEASE);
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < x.nranks; ++r) {
        if (r == x.rank) continue;
        unsigned char* c = x.cell(r, k);
        for (unsigned spins = 0; __atomic_load_n((unsigned long long*)c, __ATOMIC_ACQUIRE) != k; ++spins) {
            if ((spins & 1023) == 1023) {
                sched_yield();
                struct timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (t1.tv_sec - t0.tv_sec > 60)
                    return fail(CRIM_ERR_STATE, "split search: rank %d did not publish iteration %llu within 60 s", r, k);
            }
        }
        memcpy(x.pinned + (size_t)r * x.share, c + 64, x.share);
        HIP_TRY(hipMemcpyAsync(d_slots + (size_t)r * x.share, x.pinned + (size_t)r * x.share, x.share,
                               hipMemcpyHostToDevice, s));
    }
    h->split_exchanges += 1;
    return CRIM_OK;
}

// Captures `iters` split iterations (no host-staged exchange) into a graph.  RCCL calls are capturable;
// should the runtime refuse, the caller falls back to eager launches and remembers why.
static int build_split_graph(crim_engine* h, const Params& p, hipStream_t s, int iters, GraphSet* out) {
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = CRIM_OK;
    const int period = compact_period(p);
    for (int i = 0; i < iters && rc == CRIM_OK; ++i) {
        rc = launch_split_iteration(h, p, s, true, true);
        if ((i + 1) % period == 0) launch_compact(p, s);
    }
    const hipError_t ce = hipStreamEndCapture(s, &graph);
    if (rc != CRIM_OK || ce != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (rc == CRIM_OK) fail(CRIM_ERR_HIP, "stream capture of the split iteration: %s", hipGetErrorString(ce));
        return CRIM_ERR_HIP;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(CRIM_ERR_HIP, "hipGraphInstantiate (split iteration): %s", hipGetErrorString(e));
    out->exec = exec;
    out->iters = iters;
    return CRIM_OK;
}

// Fill loop with the search split into parts (one image).
static int run_split(crim_engine* h, long max_iters) {
    Lane& ln = h->lanes[0];
    Params p = ln.prm;
    const bool eager_flag = (h->flags & CRIM_FLAG_EAGER) != 0;
    const bool want_graphs = !eager_flag && !h->hx && !h->split_capture_failed && !getenv("CRIM_SPLIT_EAGER");
    if (want_graphs && h->split_graphs.empty()) {
        static const int sizes[3] = {64, 8, 1};
        for (int sz : sizes) {
            GraphSet g;
            if (build_split_graph(h, p, ln.stream, sz, &g) != CRIM_OK) {
                h->split_capture_failed = true;   // crim_last_error() keeps the reason
                for (auto& q : h->split_graphs)
                    if (q.exec) (void)hipGraphExecDestroy(q.exec);
                h->split_graphs.clear();
                break;
            }
            h->split_graphs.push_back(g);
        }
    }
    const bool graphs = want_graphs && !h->split_graphs.empty();
    long done = 0, last = 0;
    int prev_remaining = -1;
    for (;;) {
        HIP_TRY(hipMemcpyAsync(h->h_state, p.st, sizeof(ImgState), hipMemcpyDeviceToHost, ln.stream));
        HIP_TRY(hipStreamSynchronize(ln.stream));
        const ImgState& st = h->h_state[0];
        if (st.stalled || !st.live) break;
        long need = std::max(1L, ((long)st.remaining + p.P * p.P - 1) / (p.P * p.P));
        // (every rank sees the same state, so every rank extrapolates to the same number of iterations)
        const int filled = prev_remaining - st.remaining;
        if (last > 0 && prev_remaining >= 0 && filled > 0)
            need = std::max(need, (long)((double)st.remaining * (double)last / (double)filled));
        prev_remaining = st.remaining;
        if (!graphs) need = std::min(need, 256L);
        if (max_iters > 0) need = std::min(need, max_iters - done);
        if (need <= 0) break;
        if (p.fused) launch_priority(p, ln.stream);   // the sequence's first priority stage (idempotent)
        if (graphs) {
            long left = need;
            for (auto& g : h->split_graphs)
                while (left >= g.iters) {
                    HIP_TRY(hipGraphLaunch(g.exec, ln.stream));
                    left -= g.iters;
                }
        } else {
            for (long it = 0; it < need; ++it) {
                if ((done + it) > 0 && (done + it) % compact_period(p) == 0) launch_compact(p, ln.stream);
                int rc = launch_split_iteration(h, p, ln.stream, true, false);
                if (rc == CRIM_OK && h->hx) rc = host_exchange(h, p, ln.stream);
                if (rc == CRIM_OK) rc = launch_split_iteration(h, p, ln.stream, false, true);
                if (rc) return rc;
            }
        }
        done += need;
        last = need;
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(h->h_state, p.st, sizeof(ImgState), hipMemcpyDeviceToHost, ln.stream));
    HIP_TRY(hipStreamSynchronize(ln.stream));
    return CRIM_OK;
}

extern "C" int crim_split_unique_id(void* out128) {
    if (!out128) return fail(CRIM_ERR_ARG, "crim_split_unique_id: null argument");
    const int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    const ncclResult_t r = g_rccl.get_unique_id(&id);
    if (r != ncclSuccess) return fail(CRIM_ERR_HIP, "ncclGetUniqueId failed (%d)", (int)r);
    memcpy(out128, &id, sizeof(id));
    return CRIM_OK;
}

extern "C" int crim_split_init(crim_t* h, int nranks, int rank, const void* unique_id128) {
    if (!h || !unique_id128) return fail(CRIM_ERR_ARG, "crim_split_init: null argument");
    if (h->prm.B != 1) return fail(CRIM_ERR_ARG, "crim_split_init: the split search handles one image per engine");
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks)
        return fail(CRIM_ERR_ARG, "crim_split_init: rank %d of %d", rank, nranks);
    const int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    if (h->comm) { (void)g_rccl.comm_destroy(h->comm); h->comm = nullptr; }
    HIP_TRY(hipDeviceSynchronize());
    (void)hipGetLastError();   // RCCL treats a stale error left by an earlier, unrelated call as its own
    const ncclResult_t r = g_rccl.comm_init_rank(&h->comm, nranks, id, rank);
    if (r != ncclSuccess)
        return fail(CRIM_ERR_HIP, "ncclCommInitRank: %s", g_rccl.error_string ? g_rccl.error_string(r) : "error");
    close_host_exchange(h);
    free_graphs(h);
    h->prepared = false;
    h->split_capture_failed = false;
    h->split_parts = nranks;
    h->split_rank = rank;
    return CRIM_OK;
}

extern "C" int crim_split_init_host(crim_t* h, int nranks, int rank, const char* shared_file) {
    if (!h || !shared_file) return fail(CRIM_ERR_ARG, "crim_split_init_host: null argument");
    if (h->prm.B != 1) return fail(CRIM_ERR_ARG, "crim_split_init_host: the split search handles one image per engine");
    if (nranks < 1 || nranks > 63 || rank < 0 || rank >= nranks)
        return fail(CRIM_ERR_ARG, "crim_split_init_host: rank %d of %d", rank, nranks);
    HIP_TRY(hipSetDevice(h->device));
    if (h->comm && g_rccl.comm_destroy) { (void)g_rccl.comm_destroy(h->comm); h->comm = nullptr; }
    close_host_exchange(h);
    HostExchange* x = new (std::nothrow) HostExchange();
    if (!x) return fail(CRIM_ERR_NOMEM, "crim_split_init_host: out of host memory");
    h->hx = x;
    x->nranks = nranks;
    x->rank = rank;
    x->share = (size_t)split_rows_per_part(h->prm, nranks) * h->prm.snx * sizeof(SEntry);
    x->cell_bytes = 64 + ((x->share + 63) & ~(size_t)63);
    x->bytes = (size_t)nranks * 2 * x->cell_bytes;
    x->fd = open(shared_file, O_RDWR | O_CREAT, 0600);
    if (x->fd < 0 || ftruncate(x->fd, (off_t)x->bytes) != 0) {
        close_host_exchange(h);
        return fail(CRIM_ERR_STATE, "crim_split_init_host: cannot open or size %s", shared_file);
    }
    void* m = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
    if (m == MAP_FAILED) {
        close_host_exchange(h);
        return fail(CRIM_ERR_STATE, "crim_split_init_host: cannot map %s", shared_file);
    }
    x->map = (unsigned char*)m;
    if (hipHostMalloc((void**)&x->pinned, (size_t)nranks * x->share, hipHostMallocDefault) != hipSuccess) {
        x->pinned = nullptr;
        close_host_exchange(h);
        return fail(CRIM_ERR_NOMEM, "crim_split_init_host: page-locked staging");
    }
    free_graphs(h);
    h->prepared = false;
    h->split_parts = nranks;
    h->split_rank = rank;
    h->split_exchanges = 0;
    return CRIM_OK;
}

// How the split loop ran: *graph_captured = 1 when its iterations are replayed from captured graphs (0:
// eager launches, because a host-staged exchange sits inside every iteration, the engine is eager, or
// the runtime refused the capture), *host_exchanges = records exchanged through host memory so far.
extern "C" int crim_split_status(crim_t* h, int* graph_captured, long* host_exchanges) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_split_status: null handle");
    if (graph_captured) *graph_captured = h->split_graphs.empty() ? 0 : 1;
    if (host_exchanges) *host_exchanges = h->split_exchanges;
    return CRIM_OK;
}

// What the communicator itself reports (not what crim_split_init was told).
extern "C" int crim_split_info(crim_t* h, int* nranks, int* rank) {
    if (!h || !nranks || !rank) return fail(CRIM_ERR_ARG, "crim_split_info: null argument");
    if (!h->comm) return fail(CRIM_ERR_STATE, "crim_split_info: crim_split_init has not been called");
    if (!g_rccl.comm_count || !g_rccl.comm_user_rank)
        return fail(CRIM_ERR_STATE, "crim_split_info: librccl.so lacks ncclCommCount / ncclCommUserRank");
    if (g_rccl.comm_count(h->comm, nranks) != ncclSuccess || g_rccl.comm_user_rank(h->comm, rank) != ncclSuccess)
        return fail(CRIM_ERR_HIP, "crim_split_info: RCCL query failed");
    return CRIM_OK;
}

extern "C" int crim_split_virtual(crim_t* h, int nparts) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_split_virtual: null handle");
    if (h->prm.B != 1) return fail(CRIM_ERR_ARG, "crim_split_virtual: the split search handles one image per engine");
    if (nparts < 0 || nparts > 64) return fail(CRIM_ERR_ARG, "crim_split_virtual: %d parts", nparts);
    close_host_exchange(h);
    if (h->split_parts != nparts || h->split_rank != -1) {
        free_graphs(h);
        h->prepared = false;
    }
    h->split_parts = nparts;
    h->split_rank = -1;
    return CRIM_OK;
}

extern "C" int crim_run(crim_t* h, long max_iters, long* iters_done, long* remaining) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_run: null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->prepared) {
        const int rc = prepare(h);
        if (rc) return rc;
    }
    Params& p = h->prm;
    const bool profile = (h->flags & CRIM_FLAG_PROFILE) != 0;
    const bool eager = (h->flags & CRIM_FLAG_EAGER) != 0;
    const bool solo = (h->flags & CRIM_FLAG_PROFILE_SOLO) != 0;
    for (auto& ln : h->lanes) { ln.done = 0; ln.last = 0; }
    h->prev_remaining.assign((size_t)p.B, -1);
    int rc = CRIM_OK;
    if (h->split_parts > 0) {
        const int r2 = run_split(h, max_iters);
        if (r2) return r2;
        if (h->h_state[0].stalled) rc = CRIM_ERR_NO_PROGRESS;
    }
    for (; h->split_parts == 0;) {
        // every lane is idle here
        HIP_TRY(hipMemcpyAsync(h->h_state, p.st, (size_t)p.B * sizeof(ImgState), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int i = 0; i < p.B; ++i)
            if (h->h_state[i].stalled) rc = CRIM_ERR_NO_PROGRESS;
        if (rc) break;
        bool any = false;
        for (auto& ln : h->lanes) {
            const Params& lp = ln.prm;
            long need = 0, live = 0;
            double est = 0.0;   // iterations the slowest image still needs at the rate of the last round
            for (int i = lp.b0; i < lp.b0 + lp.B; ++i) {
                const ImgState& s = h->h_state[i];
                if (!s.live) { h->prev_remaining[i] = 0; continue; }
                ++live;
                // every iteration fills at most P*P pixels: a lower bound on what is still needed
                const long lb = std::max(1L, ((long)s.remaining + p.P * p.P - 1) / (p.P * p.P));
                need = std::max(need, lb);
                const int filled = h->prev_remaining[i] - s.remaining;
                if (ln.last > 0 && h->prev_remaining[i] >= 0 && filled > 0)
                    est = std::max(est, (double)s.remaining * (double)ln.last / (double)filled);
                h->prev_remaining[i] = s.remaining;
            }
            // Rounds end in a host round trip with the lanes drained, so they should be few: the
            // extrapolated need is launched at once (an image that finishes early only makes the
            // rest of its iterations empty launches); the lower bound keeps the first round honest.
            // (measured: extrapolation factors 0.6 ... 1.5 are within 0.5 % of each other)
            if (!profile && !eager) need = std::max(need, (long)est);
            if (live == 0) continue;
            if (max_iters > 0) need = std::min(need, max_iters - ln.done);
            if (need <= 0) continue;
            any = true;
            if (profile && !solo) {
                // as the loop runs in production: the lanes overlap, every search launch bracketed
                for (long i = 0; i < need; ++i) {
                    if ((ln.done + i) > 0 && (ln.done + i) % compact_period(p) == 0) launch_compact(lp, ln.stream);
                    if (2 * h->ev_used == (int)h->ev.size()) {
                        hipEvent_t a = nullptr, b = nullptr;
                        HIP_TRY(hipEventCreate(&a));
                        HIP_TRY(hipEventCreate(&b));
                        h->ev.push_back(a);
                        h->ev.push_back(b);
                        h->ev_evals.push_back(0.0);
                    }
                    h->ev_evals[h->ev_used] = 0.0;   // counted from the iteration counters at the end
                    launch_iteration(lp, ln.stream, h->ev[2 * h->ev_used], h->ev[2 * h->ev_used + 1]);
                    ++h->ev_used;
                }
                ln.done += need;
            } else if (profile) {
                // one iteration per round so that the number of live images per launch is known
                const double evals = (double)live * (double)std::max(p.H - p.P + 1, 0) * (double)std::max(p.W - p.P + 1, 0);
                h->ev_evals[h->ev_used] = evals;
                if (ln.done > 0 && ln.done % compact_period(p) == 0) launch_compact(lp, ln.stream);
                launch_iteration(lp, ln.stream, h->ev[2 * h->ev_used], h->ev[2 * h->ev_used + 1]);
                ++h->ev_used;
                // the bracketed launch must not share the GPU with another lane's kernels
                HIP_TRY(hipStreamSynchronize(ln.stream));
                if (h->ev_used == (int)h->ev_evals.size()) {
                    const int r2 = flush_profile(h);
                    if (r2) return r2;
                }
                ln.done += 1;
            } else if (eager) {
                for (long i = 0; i < need; ++i) {
                    if ((ln.done + i) > 0 && (ln.done + i) % compact_period(p) == 0) launch_compact(lp, ln.stream);
                    launch_iteration(lp, ln.stream);
                }
                ln.done += need;
            } else {
                // two-launch iterations: the sequence starts with the priority stage of its first iteration
                if (lp.fused) launch_priority(lp, ln.stream);
                long left = need;
                for (auto& g : ln.graphs)
                    while (left >= g.iters) {
                        HIP_TRY(hipGraphLaunch(g.exec, ln.stream));
                        left -= g.iters;
                    }
                ln.done += need;
            }
            ln.last = need;
        }
        HIP_TRY(hipGetLastError());
        if (!any) break;
        for (auto& ln : h->lanes) HIP_TRY(hipStreamSynchronize(ln.stream));
        if (profile && !solo) {
            const int r2 = flush_profile(h);
            if (r2) return r2;
        }
    }
    if (profile) {
        const int r2 = flush_profile(h);
        if (r2) return r2;
    }
    long it = 0, rem = 0;
    double searched = 0.0;
    for (int i = 0; i < p.B; ++i) {
        it = std::max(it, (long)h->h_state[i].iter);
        rem += h->h_state[i].remaining;
        searched += (double)h->h_state[i].iter;
    }
    // an image is searched once in every iteration it takes part in
    if (profile && !solo && h->split_parts == 0)
        h->prof_evals = searched * (double)std::max(p.H - p.P + 1, 0) * (double)std::max(p.W - p.P + 1, 0);
    if (iters_done) *iters_done = it;
    if (remaining) *remaining = rem;
    if (rc == CRIM_ERR_NO_PROGRESS)
        return fail(rc, "crim_run: an iteration filled no pixel while %ld remain (the reference loop would not terminate)", rem);
    return CRIM_OK;
}

extern "C" int crim_download_batch(crim_t* h, int first, int count, uint8_t* bgr_out,
                                   double* conf_out, uint8_t* mask_out) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_download: null handle");
    const Params& p = h->prm;
    if (first < 0 || count < 1 || first + count > p.B)
        return fail(CRIM_ERR_ARG, "crim_download: images [%d,%d) outside batch of %d", first, first + count, p.B);
    HIP_TRY(hipSetDevice(h->device));
    const size_t HW = (size_t)p.H * p.W;
    HIP_TRY(hipMemcpyAsync(h->h_state, p.st, (size_t)p.B * sizeof(ImgState), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bgr_out)
        HIP_TRY(hipMemcpyAsync(bgr_out, p.bgr + (size_t)first * HW * 3, (size_t)count * HW * 3,
                               hipMemcpyDeviceToHost, h->stream));
    if (mask_out)
        HIP_TRY(hipMemcpyAsync(mask_out, p.mask + (size_t)first * HW, (size_t)count * HW,
                               hipMemcpyDeviceToHost, h->stream));
    if (conf_out)
        for (int i = 0; i < count; ++i) {
            const int cur = h->h_state[first + i].cur;
            HIP_TRY(hipMemcpyAsync(conf_out + (size_t)i * HW,
                                   p.conf + ((size_t)(first + i) * 2 + cur) * HW, HW * sizeof(double),
                                   hipMemcpyDeviceToHost, h->stream));
        }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CRIM_OK;
}

extern "C" int crim_download(crim_t* h, uint8_t* bgr_out, double* conf_out, uint8_t* mask_out) {
    return crim_download_batch(h, 0, 1, bgr_out, conf_out, mask_out);
}

extern "C" int crim_download_device(crim_t* h, int first, int count, void* d_bgr_out) {
    if (!h || !d_bgr_out) return fail(CRIM_ERR_ARG, "crim_download_device: null argument");
    const Params& p = h->prm;
    if (first < 0 || count < 1 || first + count > p.B)
        return fail(CRIM_ERR_ARG, "crim_download_device: images [%d,%d) outside batch of %d", first, first + count, p.B);
    HIP_TRY(hipSetDevice(h->device));
    const size_t HW = (size_t)p.H * p.W;
    HIP_TRY(hipMemcpyAsync(d_bgr_out, p.bgr + (size_t)first * HW * 3, (size_t)count * HW * 3,
                           hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CRIM_OK;
}

extern "C" int crim_iterations(crim_t* h, int index, long* n) {
    if (!h || !n) return fail(CRIM_ERR_ARG, "crim_iterations: null argument");
    const Params& p = h->prm;
    if (index < 0 || index >= p.B) return fail(CRIM_ERR_ARG, "crim_iterations: index %d of %d", index, p.B);
    HIP_TRY(hipSetDevice(h->device));
    ImgState s;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(&s, p.st + index, sizeof(s), hipMemcpyDeviceToHost));
    *n = s.iter;
    return CRIM_OK;
}

extern "C" int crim_trace(crim_t* h, int index, crim_iter_rec* buf, long cap, long* n) {
    if (!h || !n) return fail(CRIM_ERR_ARG, "crim_trace: null argument");
    const Params& p = h->prm;
    if (index < 0 || index >= p.B) return fail(CRIM_ERR_ARG, "crim_trace: index %d of %d", index, p.B);
    long iters = 0;
    const int rc = crim_iterations(h, index, &iters);
    if (rc) return rc;
    const long have = std::min(iters, p.trace_cap);
    *n = have;
    const long take = std::min(have, cap);
    if (buf && take > 0)
        HIP_TRY(hipMemcpy(buf, p.trace + (size_t)index * p.trace_cap, (size_t)take * sizeof(crim_iter_rec),
                          hipMemcpyDeviceToHost));
    return CRIM_OK;
}

extern "C" int crim_iteration_stamps(crim_t* h, int index, long long* out3, long cap, long* n) {
    if (!h || !n) return fail(CRIM_ERR_ARG, "crim_iteration_stamps: null argument");
    const Params& p = h->prm;
    if (index < 0 || index >= p.B) return fail(CRIM_ERR_ARG, "crim_iteration_stamps: index %d of %d", index, p.B);
    long iters = 0;
    const int rc = crim_iterations(h, index, &iters);
    if (rc) return rc;
    const long have = p.stime ? std::min(iters, p.trace_cap) : 0;
    *n = have;
    const long take = std::min(have, cap);
    if (out3 && take > 0)
        HIP_TRY(hipMemcpy(out3, p.stime + (size_t)index * p.trace_cap * 3, (size_t)take * 3 * sizeof(long long),
                          hipMemcpyDeviceToHost));
    return CRIM_OK;
}

extern "C" int crim_search_times(crim_t* h, int index, float* us_out, long cap, long* n) {
    if (!h || !n) return fail(CRIM_ERR_ARG, "crim_search_times: null argument");
    long have = 0;
    int rc = crim_iteration_stamps(h, index, nullptr, 0, &have);
    if (rc) return rc;
    *n = have;
    const long take = std::min(have, cap);
    if (us_out && take > 0) {
        std::vector<long long> st((size_t)take * 3);
        rc = crim_iteration_stamps(h, index, st.data(), take, &have);
        if (rc) return rc;
        for (long i = 0; i < take; ++i) us_out[i] = (float)(st[3 * i + 2] - st[3 * i + 1]) * 0.01f;
    }
    return CRIM_OK;
}

extern "C" int crim_profile(crim_t* h, double* search_ms, long* search_launches, double* evals) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_profile: null handle");
    if (!(h->flags & CRIM_FLAG_PROFILE)) return fail(CRIM_ERR_STATE, "crim_profile: engine was not created with CRIM_FLAG_PROFILE");
    if (search_ms) *search_ms = h->prof_ms;
    if (search_launches) *search_launches = h->prof_launches;
    if (evals) *evals = h->prof_evals;
    return CRIM_OK;
}

// Diagnostic: copies the 64 stamp slots (see STAMP in crim_kernels.h); zeros unless the library
// was built with -DCRIM_STAMPS.  Not part of the public header.
#ifdef CRIM_STAMPS
// Diagnostic build only: the per-block entry / end records of the last pipelined search launch.
extern "C" int crim_debug_block_stamps(crim_t* h, long long* out8192) {
    if (!h || !out8192) return fail(CRIM_ERR_ARG, "crim_debug_block_stamps: null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out8192, h->prm.dbg + 64, 8192 * sizeof(long long), hipMemcpyDeviceToHost));
    return CRIM_OK;
}
#endif

extern "C" int crim_debug_stamps(crim_t* h, long long* out64) {
    if (!h || !out64) return fail(CRIM_ERR_ARG, "crim_debug_stamps: null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out64, h->prm.dbg, 64 * sizeof(long long), hipMemcpyDeviceToHost));
    return CRIM_OK;
}

// Diagnostic: raw copy of image `index`'s loop state (16 ints: iter, live, cur, remaining, tcount,
// chroma, pcount, scount, stalled, bound, rebuilds, ...).  Not part of the public header.
extern "C" int crim_debug_state(crim_t* h, int index, int* out16) {
    if (!h || !out16) return fail(CRIM_ERR_ARG, "crim_debug_state: null argument");
    if (index < 0 || index >= h->prm.B) return fail(CRIM_ERR_ARG, "crim_debug_state: index %d of %d", index, h->prm.B);
    static_assert(sizeof(ImgState) == 16 * sizeof(int), "ImgState is 16 ints");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out16, h->prm.st + index, sizeof(ImgState), hipMemcpyDeviceToHost));
    return CRIM_OK;
}

// Diagnostic: pads image `index`'s tile list to its capacity with copies of its first entry (as
// stale duplicates would), so that the next appended tile finds the list full.  Not part of the
// public header.
extern "C" int crim_debug_fill_tile_list(crim_t* h, int index) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_debug_fill_tile_list: null handle");
    const Params& p = h->prm;
    if (index < 0 || index >= p.B) return fail(CRIM_ERR_ARG, "crim_debug_fill_tile_list: index %d of %d", index, p.B);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ImgState st;
    HIP_TRY(hipMemcpy(&st, p.st + index, sizeof(st), hipMemcpyDeviceToHost));
    std::vector<int> tl((size_t)p.tcap);
    int* d_tl = p.tlist + ((size_t)index * 2 + st.tlsel) * p.tcap;
    HIP_TRY(hipMemcpy(tl.data(), d_tl, tl.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (st.tcount == 0 || st.tcount > (unsigned)p.tcap) return fail(CRIM_ERR_STATE, "crim_debug_fill_tile_list: list holds %u entries", st.tcount);
    for (int i = (int)st.tcount; i < p.tcap; ++i) tl[i] = tl[0];
    st.tcount = (unsigned)p.tcap;
    HIP_TRY(hipMemcpy(d_tl, tl.data(), tl.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.st + index, &st, sizeof(st), hipMemcpyHostToDevice));
    return CRIM_OK;
}

// Diagnostic: the cached data terms of image `index` (H*W doubles; only entries of pixels that were
// on the front when the priority stage last evaluated their tile mean anything).  Not part of the
// public header.
extern "C" int crim_debug_dterm(crim_t* h, int index, double* out) {
    if (!h || !out) return fail(CRIM_ERR_ARG, "crim_debug_dterm: null argument");
    const Params& p = h->prm;
    if (index < 0 || index >= p.B) return fail(CRIM_ERR_ARG, "crim_debug_dterm: index %d of %d", index, p.B);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& ln : h->lanes)
        if (ln.stream) HIP_TRY(hipStreamSynchronize(ln.stream));
    const size_t HW = (size_t)p.H * p.W;
    HIP_TRY(hipMemcpy(out, p.dterm + (size_t)index * HW, HW * sizeof(double), hipMemcpyDeviceToHost));
    return CRIM_OK;
}

extern "C" int crim_peek_fields(crim_t* h, int index, uint8_t* front_out, double* data_out,
                                double* priority_out) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_peek_fields: null handle");
    const Params& p = h->prm;
    if (index < 0 || index >= p.B) return fail(CRIM_ERR_ARG, "crim_peek_fields: index %d of %d", index, p.B);
    if (!h->uploaded[index]) return fail(CRIM_ERR_STATE, "crim_peek_fields: image %d was never uploaded", index);
    HIP_TRY(hipSetDevice(h->device));
    const size_t HW = (size_t)p.H * p.W;
    if (!h->d_peek) HIP_TRY(hipMalloc(&h->d_peek, HW * 17));
    double* d_data = (double*)h->d_peek;
    double* d_prio = d_data + HW;
    uint8_t* d_front = (uint8_t*)(d_prio + HW);
    hipLaunchKernelGGL(k_fields, dim3(p.tnx, (p.H + 1) / 2), dim3(256), 0, h->stream, p, index, d_front, d_data, d_prio);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (front_out) HIP_TRY(hipMemcpy(front_out, d_front, HW, hipMemcpyDeviceToHost));
    if (data_out) HIP_TRY(hipMemcpy(data_out, d_data, HW * sizeof(double), hipMemcpyDeviceToHost));
    if (priority_out) HIP_TRY(hipMemcpy(priority_out, d_prio, HW * sizeof(double), hipMemcpyDeviceToHost));
    return CRIM_OK;
}

// ------------------------------------------------------------------------------------------
// One-call inpaint that allocates nothing
// ------------------------------------------------------------------------------------------
// For callers that cannot allocate device memory (tight serving loops, harnesses that forbid it):
// the whole engine state of ONE image -- working image, search planes, mask, grey, gradient,
// confidence ping-pong, window validity, tile lists, loop state, target descriptor, the per-wave
// and per-block partial results of the priority and search stages, Lab tables -- lives in a
// file-scope device array instead of hipMalloc'ed buffers.  Every GPU holds its own copy of the
// array (at its own address) until the process ends.  The array keeps the last call's contents:
// whatever must start from a known value is cleared or rewritten at the top of each call.

static __device__ __align__(256) unsigned char scratch[CRIM_NOALLOC_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex[kMaxDevices];       // calls on one GPU share that GPU's array: one at a time
unsigned char* g_scratch_ptr[kMaxDevices];     // device address of `scratch` on each GPU

// Address of this GPU's copy of `scratch` (looked up once per device, with that device current).
// Host code never uses the array's name for anything else: the name is a host-side stand-in, not
// a device address.
int scratch_base(int device, unsigned char** out) {
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(CRIM_ERR_HIP, "crim_inpaint_noalloc: hipGetSymbolAddress(scratch) on device %d: %s",
                        device, hipGetErrorString(e));
        g_scratch_ptr[device] = (unsigned char*)ptr;
    }
    *out = g_scratch_ptr[device];
    return CRIM_OK;
}

// Hands out 256-byte aligned pieces of the array.
struct Carver {
    unsigned char* base;
    size_t used = 0;
    template <typename T>
    void take(T** ptr, size_t bytes) {
        *ptr = (T*)(base + used);
        used += (bytes + 255) & ~(size_t)255;
    }
};
}  // namespace

// Lays the engine state of one H x W image out in the array at `base` (null: sizes only); returns the
// bytes used.
struct NoallocLayout {
    Params p;
    uint16_t *d_gamma = nullptr, *d_cbrt = nullptr;
    size_t planes_bytes = 0, invalid_bytes = 0, invbits_bytes = 0, used = 0;
};
static void noalloc_layout(NoallocLayout& L, int H, int W, int patch_size, unsigned char* base) {
    Params& p = L.p;
    init_geometry(p, 1, H, W, patch_size, 0);
    const size_t HW = (size_t)H * W;
    const LabTables& t = lab_tables();
    memcpy(p.coef, t.coef, sizeof(p.coef));
    L.planes_bytes = (size_t)CRIM_NPL * H * p.Wp + (size_t)64 * p.Wp;
    L.invalid_bytes = (size_t)H * p.Wi;
    L.invbits_bytes = (size_t)H * p.Wb * sizeof(unsigned long long);
    Carver c{base};
    c.take(&p.bgr, HW * 3);
    c.take(&p.planes, L.planes_bytes);
    c.take(&p.mask, HW);
    c.take(&p.grey, HW);
    c.take(&p.grad, HW * sizeof(int));
    c.take(&p.conf, 2 * HW * sizeof(double));
    c.take(&p.dterm, HW * sizeof(double));
    c.take(&p.invalid, L.invalid_bytes);
    c.take(&p.invbits, L.invbits_bytes);
    c.take(&p.tfront, (size_t)p.tnx * p.tny * sizeof(int));
    c.take(&p.tlist, (size_t)2 * p.tcap * sizeof(int));
    c.take(&p.st, sizeof(ImgState));
    c.take(&p.desc, sizeof(Desc));
    c.take(&p.plist, (size_t)p.pgrid * sizeof(PEntry));
    c.take(&p.slist, (size_t)p.snx * p.srows * sizeof(SEntry));
    c.take(&p.dbg, kDbgSlots * sizeof(long long));
    c.take(&L.d_gamma, sizeof(t.gamma));
    c.take(&L.d_cbrt, sizeof(t.cbrt_));
    L.used = c.used;
}

extern "C" size_t crim_noalloc_bytes_needed(int H, int W, int patch_size) {
    if (H < 1 || W < 1 || patch_size < 1 || patch_size > CRIM_MAX_PATCH) return 0;
    NoallocLayout L;
    noalloc_layout(L, H, W, patch_size, nullptr);
    return L.used;
}

extern "C" int crim_inpaint_noalloc(int H, int W, int patch_size, int device, const void* d_bgr,
                                    const void* d_mask, void* d_bgr_out, long* iters_done) {
    if (!d_bgr || !d_mask || !d_bgr_out) return fail(CRIM_ERR_ARG, "crim_inpaint_noalloc: null argument");
    if (H < 1 || W < 1) return fail(CRIM_ERR_ARG, "crim_inpaint_noalloc: bad shape %d x %d", H, W);
    if (patch_size < 1 || patch_size > CRIM_MAX_PATCH)
        return fail(CRIM_ERR_ARG, "crim_inpaint_noalloc: patch_size %d outside 1..%d", patch_size, CRIM_MAX_PATCH);
    if ((long)H * W > (long)CRIM_NOALLOC_MAX_PIXELS)
        return fail(CRIM_ERR_ARG, "crim_inpaint_noalloc: %d x %d = %ld pixels exceed the limit of %d "
                    "(use crim_create / crim_run for larger images)", H, W, (long)H * W, CRIM_NOALLOC_MAX_PIXELS);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || device >= kMaxDevices)
        return fail(CRIM_ERR_ARG, "crim_inpaint_noalloc: device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_scratch_mutex[device]);
    unsigned char* base = nullptr;
    {
        const int rc = scratch_base(device, &base);
        if (rc) return rc;
    }
    NoallocLayout lay;
    noalloc_layout(lay, H, W, patch_size, base);
    Params& p = lay.p;
    const size_t HW = (size_t)H * W;
    const LabTables& t = lab_tables();
    const size_t planes_bytes = lay.planes_bytes, invalid_bytes = lay.invalid_bytes, invbits_bytes = lay.invbits_bytes;
    uint16_t *d_gamma = lay.d_gamma, *d_cbrt = lay.d_cbrt;
    struct { size_t used; } c{lay.used};
    if (c.used > CRIM_NOALLOC_BYTES)
        return fail(CRIM_ERR_ARG, "crim_inpaint_noalloc: a %d x %d image needs %zu bytes of state, the static "
                    "buffer holds %zu (rows narrower than 32 pixels are padded heavily; use crim_create / crim_run)",
                    H, W, c.used, (size_t)CRIM_NOALLOC_BYTES);
    p.gamma_tab = d_gamma;
    p.cbrt_tab = d_cbrt;
    p.trace = nullptr;       // no per-iteration records
    const hipStream_t s = nullptr;   // the default stream: creating one would allocate
    // what the kernels expect to find zeroed / preset (the array still holds the previous call's state)
    HIP_TRY(hipMemsetAsync(p.planes, 0, planes_bytes, s));
    HIP_TRY(hipMemsetAsync(p.invalid, 1, invalid_bytes, s));
    HIP_TRY(hipMemsetAsync(p.invbits, 0xff, invbits_bytes, s));
    HIP_TRY(hipMemsetAsync(p.st, 0, sizeof(ImgState), s));
    HIP_TRY(hipMemsetAsync(p.desc, 0, sizeof(Desc), s));
    HIP_TRY(hipMemsetAsync(p.tfront, 0, (size_t)p.tnx * p.tny * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(p.tlist, 0, (size_t)2 * p.tcap * sizeof(int), s));
    HIP_TRY(hipMemcpyAsync(d_gamma, t.gamma, sizeof(t.gamma), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_cbrt, t.cbrt_, sizeof(t.cbrt_), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(p.bgr, d_bgr, HW * 3, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(p.mask, d_mask, HW, hipMemcpyDeviceToDevice, s));
    launch_init(p, s, 0, 1);
    HIP_TRY(hipGetLastError());
    ImgState st;
    HIP_TRY(hipMemcpyAsync(&st, p.st, sizeof(st), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    p.chroma = st.chroma ? 1 : 0;
    // The loop of crim_run for one image, launched eagerly in rounds: a round is the number of
    // iterations the image certainly still needs (each fills at most P*P pixels), or the need
    // extrapolated from the last round's fill rate; iterations past the end are no-ops.
    long done = 0, last = 0;
    int prev_remaining = -1;
    int rc = CRIM_OK;
    for (;;) {
        if (st.stalled) { rc = CRIM_ERR_NO_PROGRESS; break; }
        if (!st.live) break;
        long need = std::max(1L, ((long)st.remaining + p.P * p.P - 1) / (p.P * p.P));
        const int filled = prev_remaining - st.remaining;
        if (last > 0 && prev_remaining >= 0 && filled > 0)
            need = std::max(need, (long)((double)st.remaining * (double)last / (double)filled));
        need = std::min(need, 4096L);
        prev_remaining = st.remaining;
        if (p.fused) launch_priority(p, s);
        for (long i = 0; i < need; ++i) {
            if ((done + i) > 0 && (done + i) % compact_period(p) == 0) launch_compact(p, s);
            if (p.fused) launch_iteration_fused(p, s);
            else launch_iteration(p, s);
        }
        done += need;
        last = need;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&st, p.st, sizeof(st), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    HIP_TRY(hipMemcpyAsync(d_bgr_out, p.bgr, HW * 3, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (iters_done) *iters_done = st.iter;
    if (rc == CRIM_ERR_NO_PROGRESS)
        return fail(rc, "crim_inpaint_noalloc: an iteration filled no pixel while %d remain "
                    "(the reference loop would not terminate)", st.remaining);
    return CRIM_OK;
}

// ------------------------------------------------------------------------------------------
// Noise-filter stage (filter_noise, /root/reference/src/utils.py:106-138)
// ------------------------------------------------------------------------------------------
#include "crim_filters.h"

namespace {

struct NfHostTables {
    uint16_t yf[512];       // LabToYF_b: (y, f(y)) per 8-bit L in units of 1/16384, float32 arithmetic
    int coef_inv[9];        // XYZ -> linear BGR times the white point, 12 bit
};

const NfHostTables& nf_host_tables() {
    static NfHostTables t;
    static std::once_flag once;
    std::call_once(once, [] {
        const int base = 1 << 14;
        for (int i = 0; i < 256; ++i) {
            int y, ify;
            if (i <= 20) {
                y = round_half_even((double)((float)(i * base * 20 * 9) / (float)(17 * 29 * 29 * 29)));
                const float a = (float)16 / (float)116, b = (float)(i * 5) / (float)(3 * 17 * 29);
                const float s = a + b;
                ify = round_half_even((double)((float)base * s));
            } else {
                const float p = (float)(i * 100 * base) / (float)(255 * 116);
                const float q = (float)(16 * base) / (float)116;
                const float fy = p + q;
                ify = round_half_even((double)fy);
                const float f2 = fy * fy;
                const float f3 = f2 * fy;
                y = round_half_even((double)(f3 / (float)(base * base)));
            }
            t.yf[2 * i] = (uint16_t)y;
            t.yf[2 * i + 1] = (uint16_t)ify;
        }
        static const double m[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991,
                                    0.041556, 0.055648, -0.204043, 1.057311};
        static const double wp[3] = {0.950456, 1.0, 1.088754};
        for (int c = 0; c < 3; ++c)          // output channel c of B,G,R uses the XYZ -> sRGB row 2 - c
            for (int j = 0; j < 3; ++j) t.coef_inv[3 * c + j] = round_half_even(4096.0 * m[3 * (2 - c) + j] * wp[j]);
    });
    return t;
}

// almost_dist2weight_ of OpenCV's FastNlMeansDenoisingInvoker for 8-bit samples: entry a is the
// integer weight of the shifted window distance a.
std::vector<int> nlm_weight_table(float h, int channels, int* shift_out) {
    const int tsq = NF_TW * NF_TW;
    int shift = 0;
    while ((1 << shift) < tsq) ++shift;
    const double mult = (double)(1 << shift) / tsq;
    const int max_dist = 255 * 255 * channels;
    const int n = (int)(max_dist / mult + 1);
    const int fpm = std::min(2147483647 / (NF_SW * NF_SW * 255), 2147483647);
    const float hh = h * h * (float)channels;
    std::vector<int> tab((size_t)n);
    for (int a = 0; a < n; ++a) {
        const double dist = a * mult;
        double w = hh > 0 ? std::exp(-dist / hh) : 1.0;
        if (std::isnan(w)) w = 1.0;
        int weight = round_half_even(fpm * w);
        if (weight < 0.001 * fpm) weight = 0;
        tab[(size_t)a] = weight;
    }
    *shift_out = shift;
    return tab;
}

struct NfScratch {
    uint8_t *l0 = nullptr, *ab0 = nullptr, *l1 = nullptr, *ab1 = nullptr;
    int *wl = nullptr, *wab = nullptr;
    uint16_t *cbrt = nullptr, *yf = nullptr;
    void release() {
        void* q[] = {l0, ab0, l1, ab1, wl, wab, cbrt, yf};
        for (void* p : q)
            if (p) (void)hipFree(p);
    }
};

// The three launches of the stage on dense device images [count][H][W][3]; in-place allowed.
int filter_noise_device(int count, int H, int W, const uint8_t* d_in, uint8_t* d_out, int median_ksize,
                        float nlm_h, int apply_sharpening, hipStream_t s) {
    if (median_ksize != 1 && median_ksize != 3 && median_ksize != 5 && median_ksize != 7)
        return fail(CRIM_ERR_ARG, "crim_filter_noise: median_ksize %d (1, 3, 5 or 7)", median_ksize);
    if (!(nlm_h >= 0.0f) || nlm_h > 1000.0f) return fail(CRIM_ERR_ARG, "crim_filter_noise: nlm_h %g", (double)nlm_h);
    const size_t HW = (size_t)H * W, n = (size_t)count;
    NfScratch sc;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    int shift_l = 0, shift_ab = 0;
    const std::vector<int> wl = nlm_weight_table(nlm_h, 1, &shift_l);
    const std::vector<int> wab = nlm_weight_table(1.0f, 2, &shift_ab);      // hColor=1 (utils.py:125)
    const LabTables& lt = lab_tables();
    const NfHostTables& ht = nf_host_tables();
    alloc((void**)&sc.l0, n * HW);
    alloc((void**)&sc.ab0, n * HW * 2);
    alloc((void**)&sc.l1, n * HW);
    alloc((void**)&sc.ab1, n * HW * 2);
    alloc((void**)&sc.wl, wl.size() * sizeof(int));
    alloc((void**)&sc.wab, wab.size() * sizeof(int));
    alloc((void**)&sc.cbrt, sizeof(lt.cbrt_));
    alloc((void**)&sc.yf, sizeof(ht.yf));
    if (e == hipSuccess) e = hipMemcpyAsync(sc.wl, wl.data(), wl.size() * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(sc.wab, wab.data(), wab.size() * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(sc.cbrt, lt.cbrt_, sizeof(lt.cbrt_), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(sc.yf, ht.yf, sizeof(ht.yf), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        sc.release();
        return fail(e == hipErrorOutOfMemory ? CRIM_ERR_NOMEM : CRIM_ERR_HIP, "crim_filter_noise: %s", hipGetErrorString(e));
    }
    NfTables t;
    t.cbrt_tab = sc.cbrt;
    t.yf_tab = sc.yf;
    memcpy(t.coef_fwd, lt.coef, sizeof(t.coef_fwd));
    memcpy(t.coef_inv, ht.coef_inv, sizeof(t.coef_inv));
    const dim3 grid((W + 15) / 16, (H + 15) / 16, count), block(256);
    switch (median_ksize) {
        case 1: hipLaunchKernelGGL(k_nf_median_lab<1>, grid, block, 0, s, d_in, sc.l0, sc.ab0, H, W, t); break;
        case 3: hipLaunchKernelGGL(k_nf_median_lab<3>, grid, block, 0, s, d_in, sc.l0, sc.ab0, H, W, t); break;
        case 5: hipLaunchKernelGGL(k_nf_median_lab<5>, grid, block, 0, s, d_in, sc.l0, sc.ab0, H, W, t); break;
        default: hipLaunchKernelGGL(k_nf_median_lab<7>, grid, block, 0, s, d_in, sc.l0, sc.ab0, H, W, t); break;
    }
    // length of the tables' non-zero prefix: all a window distance can reach with a weight
    auto prefix = [](const std::vector<int>& w) { size_t k = w.size(); while (k > 0 && w[k - 1] == 0) --k; return (int)k; };
    if (getenv("CRIM_NLM_DIRECT")) {   // the window-difference form (A/B and cross-check)
        hipLaunchKernelGGL(k_nf_nlm<1>, grid, block, 0, s, sc.l0, sc.l1, H, W, sc.wl, prefix(wl), shift_l);
        hipLaunchKernelGGL(k_nf_nlm<2>, grid, block, 0, s, sc.ab0, sc.ab1, H, W, sc.wab, prefix(wab), shift_ab);
    } else if (!getenv("CRIM_NLM_CORR") && prefix(wl) < NF_WLDS && prefix(wab) < NF_WLDS) {
        // row products shared down a column (the default)
        const dim3 gl((W + NF2_TW - 1) / NF2_TW, (H + 15) / 16, count), gab((W + NF2_TW - 1) / NF2_TW, (H + 7) / 8, count);
        hipLaunchKernelGGL((k_nf_nlm_vs<1, 8>), gl, block, 0, s, sc.l0, sc.l1, H, W, sc.wl, prefix(wl), shift_l);
        hipLaunchKernelGGL((k_nf_nlm_vs<2, 4>), gab, block, 0, s, sc.ab0, sc.ab1, H, W, sc.wab, prefix(wab), shift_ab);
    } else {
        const dim3 g2((W + NF2_TW - 1) / NF2_TW, (H + NF2_TH - 1) / NF2_TH, count);
        hipLaunchKernelGGL(k_nf_nlm_corr<1>, g2, block, 0, s, sc.l0, sc.l1, H, W, sc.wl, prefix(wl), shift_l);
        if (getenv("CRIM_NLM_CORR2")) {
            hipLaunchKernelGGL(k_nf_nlm_corr<2>, g2, block, 0, s, sc.ab0, sc.ab1, H, W, sc.wab, prefix(wab), shift_ab);
        } else {
            const char* eb = getenv("CRIM_NLM_BLK");     // output rows per thread (experiments)
            const int blk = eb ? atoi(eb) : 4;
            const dim3 g3((W + NF3_TW - 1) / NF3_TW, (H + 4 * blk - 1) / (4 * blk), count);
            if (blk == 2) hipLaunchKernelGGL(k_nf_nlm_rows<2>, g3, block, 0, s, sc.ab0, sc.ab1, H, W, sc.wab, prefix(wab), shift_ab);
            else if (blk == 3) hipLaunchKernelGGL(k_nf_nlm_rows<3>, g3, block, 0, s, sc.ab0, sc.ab1, H, W, sc.wab, prefix(wab), shift_ab);
            else hipLaunchKernelGGL(k_nf_nlm_rows<4>, g3, block, 0, s, sc.ab0, sc.ab1, H, W, sc.wab, prefix(wab), shift_ab);
        }
    }
    hipLaunchKernelGGL(k_nf_lab2bgr_sharpen, grid, block, 0, s, sc.l1, sc.ab1, d_out, H, W, apply_sharpening ? 1 : 0, t);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    sc.release();
    if (e != hipSuccess) return fail(CRIM_ERR_HIP, "crim_filter_noise: %s", hipGetErrorString(e));
    return CRIM_OK;
}

}  // namespace

extern "C" int crim_filter_noise_device(int device, int count, int H, int W, const void* d_bgr_in, void* d_bgr_out,
                                        int median_ksize, float nlm_h, int apply_sharpening) {
    if (!d_bgr_in || !d_bgr_out) return fail(CRIM_ERR_ARG, "crim_filter_noise_device: null argument");
    if (count < 1 || H < 1 || W < 1) return fail(CRIM_ERR_ARG, "crim_filter_noise_device: bad shape %d x %d x %d", count, H, W);
    HIP_TRY(hipSetDevice(device));
    return filter_noise_device(count, H, W, (const uint8_t*)d_bgr_in, (uint8_t*)d_bgr_out, median_ksize, nlm_h,
                               apply_sharpening, nullptr);
}

extern "C" int crim_filter_noise(int device, int count, int H, int W, const uint8_t* bgr_in, uint8_t* bgr_out,
                                 int median_ksize, float nlm_h, int apply_sharpening) {
    if (!bgr_in || !bgr_out) return fail(CRIM_ERR_ARG, "crim_filter_noise: null argument");
    if (count < 1 || H < 1 || W < 1) return fail(CRIM_ERR_ARG, "crim_filter_noise: bad shape %d x %d x %d", count, H, W);
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)count * H * W * 3;
    uint8_t* d = nullptr;
    const hipError_t e = hipMalloc((void**)&d, bytes);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? CRIM_ERR_NOMEM : CRIM_ERR_HIP, "crim_filter_noise: %s", hipGetErrorString(e));
    int rc = CRIM_OK;
    if (hipMemcpy(d, bgr_in, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = fail(CRIM_ERR_HIP, "crim_filter_noise: upload failed");
    if (!rc) rc = filter_noise_device(count, H, W, d, d, median_ksize, nlm_h, apply_sharpening, nullptr);
    if (!rc && hipMemcpy(bgr_out, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(CRIM_ERR_HIP, "crim_filter_noise: download failed");
    (void)hipFree(d);
    return rc;
}

// filter_noise applied in place to the working images of an engine (after crim_run): the
// inpaint -> denoise order of process_image (utils.py:216-220) without leaving the device.
extern "C" int crim_filter_noise_engine(crim_t* h, int median_ksize, float nlm_h, int apply_sharpening) {
    if (!h) return fail(CRIM_ERR_ARG, "crim_filter_noise_engine: null handle");
    HIP_TRY(hipSetDevice(h->device));
    for (auto& ln : h->lanes) HIP_TRY(hipStreamSynchronize(ln.stream));
    const Params& p = h->prm;
    return filter_noise_device(p.B, p.H, p.W, p.bgr, p.bgr, median_ksize, nlm_h, apply_sharpening, h->stream);
}
