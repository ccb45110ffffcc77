/*
 * C ABI of libcriminisi_hip.so: a device-resident Criminisi exemplar-inpainting engine for
 * AMD MI355X (gfx950).
 *
 * The reference (Theosdoor/Image-Processing) is pure Python and has no FFI; its only interface
 * for this path is the class Criminisi_Inpainter(image, mask, patch_size, verbose,
 * show_progress) with .inpaint() (src/criminisi_inpainter.py:7,18,40-88), constructed once per
 * image by src/utils.py:102-103.  The entry points below are what a ctypes binding for that
 * class needs: create an engine for one image shape, hand over image + mask, run the fill loop
 * to completion on the GPU, read back the result and (for parity checks) the per-iteration
 * decisions.  Each declaration cites the reference lines it replaces.
 *
 * Conventions: every function returns CRIM_OK (0) or a positive error code and never aborts;
 * crim_last_error() gives the message for the calling thread's last failure.  A handle may be
 * used from one host thread at a time; different handles may be used concurrently.  Images
 * are uint8 H x W x 3 in B,G,R order; masks are uint8 H x W, nonzero = pixel to fill.
 */
#ifndef CRIMINISI_HIP_H
#define CRIMINISI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crim_engine crim_t;

/* One record per executed iteration of the fill loop (src/criminisi_inpainter.py:51-83):
 * the target pixel chosen at :62, the top-left corner of the source patch chosen at :65
 * (:264-265), the fp64 cost of that patch (:247-262) and the number of masked pixels left
 * after the update (:74). */
typedef struct crim_iter_rec {
    int32_t tr, tc;
    int32_t sy, sx;
    double cost;
    int64_t remaining;
} crim_iter_rec;

enum {
    CRIM_OK = 0,
    CRIM_ERR_ARG = 1,         /* bad argument (null pointer, size out of range, index) */
    CRIM_ERR_HIP = 2,         /* a HIP runtime call failed; see crim_last_error() */
    CRIM_ERR_STATE = 3,       /* call made in the wrong order (run before upload, ...) */
    CRIM_ERR_NO_PROGRESS = 4, /* an iteration filled nothing while pixels remain: the reference
                                 loop (:51-83) would never terminate on this input */
    CRIM_ERR_NOMEM = 5
};

/* crim_create* flags */
enum {
    CRIM_FLAG_EAGER = 1u,          /* launch kernels one by one instead of replaying a hipGraph */
    CRIM_FLAG_PROFILE = 2u,        /* bracket every search launch with HIP events (implies eager);
                                      the lanes of a batch overlap as they do in production */
    CRIM_FLAG_GENERIC_SEARCH = 4u, /* use the unspecialised search kernel for every patch size */
    CRIM_FLAG_DOT4_SEARCH = 8u,    /* use the vector dot-product search kernel instead of the
                                      matrix-core one (patch sizes 3..16) */
    CRIM_FLAG_PROFILE_SOLO = 16u,  /* as CRIM_FLAG_PROFILE, but every bracketed search launch has the
                                      GPU to itself (the stream is drained after each iteration) */
    CRIM_FLAG_NO_TRACE = 32u,      /* keep no per-iteration records (crim_trace then returns none):
                                      saves 32 bytes per image and possible iteration */
    CRIM_FLAG_ONE_LANE = 64u       /* the whole batch advances on one stream (for a caller that runs
                                      several engines side by side, e.g. to overlap their transfers) */
};

/* Largest patch_size accepted.  The reference takes any integer (:18); the bit-exact regime
 * of its float32 SSD is patch_size <= 16 (per-channel sums stay below 2^24). */
#define CRIM_MAX_PATCH 31

int crim_device_count(void);
const char* crim_last_error(void);

/* Criminisi_Inpainter.__init__ (:18-38) for one image of H x W: allocates all device state. */
int crim_create(int H, int W, int patch_size, int device, unsigned flags, crim_t** out);
/* Same for `batch` independent images of one shape that advance in lock step; replaces the
 * per-file loop around the constructor in scripts/main.py:44-58. */
int crim_create_batch(int batch, int H, int W, int patch_size, int device, unsigned flags,
                      crim_t** out);
void crim_destroy(crim_t* h);

/* Hands over working_image / working_mask (:29-30) and derives confidence (:35) and all
 * cached per-pixel fields.  Strides are in bytes; inputs are host memory and are not modified. */
int crim_upload(crim_t* h, const uint8_t* bgr, ptrdiff_t row_stride, const uint8_t* mask,
                ptrdiff_t mask_row_stride);
int crim_upload_batch(crim_t* h, int first, int count, const uint8_t* bgr, ptrdiff_t image_stride,
                      ptrdiff_t row_stride, const uint8_t* mask, ptrdiff_t mask_image_stride,
                      ptrdiff_t mask_row_stride);
/* Hands over the image only and builds the mask on the GPU with the recipe of the reference's
 * caller, inpaint_image (src/utils.py:79-100): grey <= threshold (10 there), a border_width (10)
 * pixel frame cleared, one dilation_size x dilation_size (4) box dilation, then {0,1}. */
int crim_upload_automask(crim_t* h, int first, int count, const uint8_t* bgr, ptrdiff_t image_stride,
                         ptrdiff_t row_stride, int threshold, int border_width, int dilation_size);
/* Same, from dense device buffers ([count][H][W][3] and [count][H][W]) on the engine's GPU.  The
 * buffers are read by kernels enqueued on the engine's stream (there is no copy at call time): leave
 * them unchanged until a later call that waits for the engine has returned (crim_run does). */
int crim_upload_device(crim_t* h, int first, int count, const void* d_bgr, const void* d_mask);

/* Page-locked host memory (hipHostMalloc / hipHostFree): uploads from and downloads into it are
 * DMA transfers that overlap with the kernels of other engines, which is how a host that streams
 * batches hides PCIe behind the fill loop.  No counterpart in the reference (its images never
 * leave host memory). */
int crim_host_alloc(size_t bytes, void** out);
int crim_host_free(void* ptr);

/* inpaint() (:40-88): runs the loop until every image's mask is empty, or until max_iters
 * iterations have been executed in this call (max_iters <= 0: no limit).  iters_done receives
 * the largest per-image iteration count so far, remaining the total masked pixels left. */
int crim_run(crim_t* h, long max_iters, long* iters_done, long* remaining);

/* Split search for ONE large image (batch 1): the block rows of window origins scored by
 * _find_source_patch (:240-264) are divided into equal parts; a search block's best window is a
 * 16-byte (cost, index) slot, the parts' slots are contiguous pieces of one array, and the
 * lexicographic minimum over all slots is the reference's arg-min.  No counterpart in the reference (it
 * is single-process).
 *   crim_split_virtual: the parts are scored one after another by this engine (checks the
 *     partitioning on one GPU; nparts 0 switches the split off).
 *   crim_split_init: this process scores part `rank` of `nranks`; every process holds a replica
 *     of the image state and the parts' slots are put together by one in-place RCCL all-gather per
 *     iteration (a few KB per rank; captured, with the iteration's kernels, into the hipGraphs the loop
 *     replays), after which every process reduces all slots exactly as after an unsplit search.
 *     unique_id128 is the 128-byte id rank 0 obtained from crim_split_unique_id and shared out of
 *     band (e.g. torch.distributed broadcast). */
int crim_split_virtual(crim_t* h, int nparts);
int crim_split_unique_id(void* out128);
int crim_split_init(crim_t* h, int nranks, int rank, const void* unique_id128);
/* Size and rank of the RCCL communicator as RCCL itself reports them (after crim_split_init). */
int crim_split_info(crim_t* h, int* nranks, int* rank);
/* The same split with the slots exchanged through host memory instead of RCCL: every rank's process
 * (all on one host) maps `shared_file` (a fresh path, the same for all ranks), publishes its share there
 * each iteration and reads the others' (device -> host, shared mapping, host -> device).  For
 * tests and for placements RCCL refuses, e.g. two ranks on ONE GPU; the host is in the loop every
 * iteration, so it is slow. */
int crim_split_init_host(crim_t* h, int nranks, int rank, const char* shared_file);
/* How the split loop runs: *graph_captured = 1 when its iterations (RCCL all-gather included) are
 * replayed from captured hipGraphs, 0 when they are launched one by one (host-staged exchange, eager
 * engine, or a runtime that refused the capture: crim_last_error() then holds the reason);
 * *host_exchanges = iterations exchanged through host memory so far.  Null outputs are skipped. */
int crim_split_status(crim_t* h, int* graph_captured, long* host_exchanges);

/* working_image (:88), confidence and working_mask after the run; null outputs are skipped. */
int crim_download(crim_t* h, uint8_t* bgr_out, double* conf_out, uint8_t* mask_out);
int crim_download_batch(crim_t* h, int first, int count, uint8_t* bgr_out, double* conf_out,
                        uint8_t* mask_out);
/* Dense device-side copy of the working images ([count][H][W][3]). */
int crim_download_device(crim_t* h, int first, int count, void* d_bgr_out);

/* The whole of crim_create + crim_upload_device + crim_run + crim_download_device + crim_destroy
 * for ONE image, without allocating anything.  No counterpart in the reference; it is for callers
 * that cannot allocate device memory while they run (tight serving loops, harnesses that forbid
 * hipMalloc): all engine state, including the partial results of the priority and search stages,
 * lives in one static device array of CRIM_NOALLOC_BYTES (52 bytes per pixel of the largest image
 * + 256 KiB = 13.25 MiB) that each GPU the library runs on holds until the process ends.  d_bgr ([H][W][3]), d_mask
 * ([H][W], nonzero = fill) and d_bgr_out ([H][W][3]) are dense device buffers on GPU `device`.
 * H * W must not exceed CRIM_NOALLOC_MAX_PIXELS (every such shape with H >= 32 and W >= 32 fits;
 * narrower or flatter images may be refused because their plane rows are padded and the planes end in
 * 64 spare rows; crim_noalloc_bytes_needed tells): CRIM_ERR_ARG otherwise.  Work is
 * issued on the default stream and the call returns when the result is in d_bgr_out.  Calls for
 * one GPU share that GPU's array: the library serialises them within a process (a lock per GPU);
 * calls from different handles of crim_create are independent of it.  No per-iteration records. */
#define CRIM_NOALLOC_MAX_PIXELS (512 * 512)
#define CRIM_NOALLOC_BYTES ((size_t)52 * CRIM_NOALLOC_MAX_PIXELS + ((size_t)1 << 18))
/* Bytes of that array a call for this shape would use (0 for a bad argument); needs no GPU. */
size_t crim_noalloc_bytes_needed(int H, int W, int patch_size);
int crim_inpaint_noalloc(int H, int W, int patch_size, int device, const void* d_bgr,
                         const void* d_mask, void* d_bgr_out, long* iters_done);

/* filter_noise(image, median_ksize, nlm_h, apply_sharpening) (src/utils.py:106-138), the stage
 * that follows the inpainter in process_image (:216-220), for `count` dense images [count][H][W][3]:
 * cv2.medianBlur(median_ksize) (:118), cv2.fastNlMeansDenoisingColored(h=nlm_h, hColor=1,
 * templateWindowSize=9, searchWindowSize=31) (:124-126) and, if apply_sharpening, the Laplacian
 * sharpening of :128-136.  median_ksize is 1, 3, 5 or 7.  Host buffers, device buffers (in place
 * allowed), or the working images of an engine after crim_run (in place, the images stay on the GPU). */
int crim_filter_noise(int device, int count, int H, int W, const uint8_t* bgr_in, uint8_t* bgr_out,
                      int median_ksize, float nlm_h, int apply_sharpening);
int crim_filter_noise_device(int device, int count, int H, int W, const void* d_bgr_in, void* d_bgr_out,
                             int median_ksize, float nlm_h, int apply_sharpening);
int crim_filter_noise_engine(crim_t* h, int median_ksize, float nlm_h, int apply_sharpening);

/* Per-iteration decisions of image `index` (parity instrument; no counterpart in the reference
 * beyond the values its loop computes at :62-74). */
int crim_trace(crim_t* h, int index, crim_iter_rec* buf, long cap, long* n);
/* Iterations executed so far by image `index`. */
int crim_iterations(crim_t* h, int index, long* n);
/* 'Time to find best' of every executed iteration of image `index` (the reference times
 * _find_source_patch per iteration with time.time(), :64-69): microseconds from the end of the target
 * selection to the start of the update on the device's 100 MHz clock, i.e. the search launch and the
 * launch boundary after it; lock-step batches share the launch, so every image of a batch sees its
 * batch's time.  Kept with the per-iteration records (none under CRIM_FLAG_NO_TRACE). */
int crim_search_times(crim_t* h, int index, float* us_out, long cap, long* n);
/* The raw stamps behind it, three per executed iteration (ticks of the device's 100 MHz clock): start of
 * the launch that selects the target, end of the selection, start of the update.  Differences between
 * consecutive iterations give the time of each launch of the loop as the device saw it (measurement
 * instrument; no counterpart in the reference). */
int crim_iteration_stamps(crim_t* h, int index, long long* out3, long cap, long* n);

/* The integer BGR->Lab tables the engine uses in place of cv2.cvtColor(..., COLOR_BGR2LAB)
 * (:225): 256-entry sRGB gamma table, 3072-entry cube-root table, 9 matrix coefficients
 * (row-major X,Y,Z; columns ordered B,G,R). */
int crim_lab_tables(uint16_t* gamma256, uint16_t* cbrt3072, int32_t* coeffs9);

/* Accumulated HIP-event time of the search kernel (needs CRIM_FLAG_PROFILE) since the last
 * upload: total milliseconds, number of launches, candidate windows scored per launch summed. */
int crim_profile(crim_t* h, double* search_ms, long* search_launches, double* evals);

/* Debug/parity: full-frame front (:103), data (:134) and priority (:112) of image `index` for
 * the iteration that would run next, computed without advancing the loop.  Null outputs are
 * skipped. */
int crim_peek_fields(crim_t* h, int index, uint8_t* front_out, double* data_out,
                     double* priority_out);

/* ---- debug / parity instruments (used by the test-suite; no counterpart in the reference) ---------- */
/* Raw copy of image `index`'s loop state: 16 ints (iter, live, cur, remaining, tile-list length, chroma,
 * two unused, stalled, pruning bound, tile-list rebuilds, tile-list buffer, ...). */
int crim_debug_state(crim_t* h, int index, int* out16);
/* Pads image `index`'s tile list to its capacity with copies of its first entry, so that the next
 * appended tile finds the list full and the in-kernel rebuild has to run. */
int crim_debug_fill_tile_list(crim_t* h, int index);
/* The cached data terms of image `index` (H*W doubles; meaningful only at pixels that were on the front
 * when the priority stage last evaluated their tile). */
int crim_debug_dterm(crim_t* h, int index, double* out);
/* The 64 named time-stamp slots of the iteration kernels (zeros unless the library was built with
 * -DCRIM_STAMPS). */
int crim_debug_stamps(crim_t* h, long long* out64);

#ifdef __cplusplus
}
#endif
#endif
