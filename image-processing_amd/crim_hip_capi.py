"""ctypes binding of libcriminisi_hip.so (C ABI: include/criminisi_hip.h).

The library is the product: if it is missing or no MI355X is visible, importing succeeds (so
that error paths can be tested on CPU) but creating an engine raises -- there is no CPU
fallback anywhere in this package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRIM_HIP_LIB points at an alternative build of the same library (diagnostic builds)
LIB_PATH = os.environ.get("CRIM_HIP_LIB") or os.path.join(_HERE, "lib", "libcriminisi_hip.so")

CRIM_OK = 0
CRIM_ERR_ARG = 1
CRIM_ERR_HIP = 2
CRIM_ERR_STATE = 3
CRIM_ERR_NO_PROGRESS = 4
CRIM_ERR_NOMEM = 5

FLAG_EAGER = 1
FLAG_PROFILE = 2
FLAG_GENERIC_SEARCH = 4
FLAG_DOT4_SEARCH = 8
FLAG_PROFILE_SOLO = 16
FLAG_NO_TRACE = 32
FLAG_ONE_LANE = 64

MAX_PATCH = 31

ITER_REC_DTYPE = np.dtype([("tr", "<i4"), ("tc", "<i4"), ("sy", "<i4"), ("sx", "<i4"),
                           ("cost", "<f8"), ("remaining", "<i8")])


class CriminisiHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libcriminisi_hip error %d: %s" % (code, message))
        self.code = code


_lib = None


def load_library():
    """Loads the shared library once; raises OSError with the expected path if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libcriminisi_hip.so not found at %s -- build it with "
                      "image-processing_amd/csrc/build.sh (needs hipcc, targets gfx950)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    pd = ctypes.c_ssize_t
    lib.crim_device_count.restype = ci
    lib.crim_device_count.argtypes = []
    lib.crim_last_error.restype = ctypes.c_char_p
    lib.crim_last_error.argtypes = []
    lib.crim_create.restype = ci
    lib.crim_create.argtypes = [ci, ci, ci, ci, ctypes.c_uint, ctypes.POINTER(vp)]
    lib.crim_create_batch.restype = ci
    lib.crim_create_batch.argtypes = [ci, ci, ci, ci, ci, ctypes.c_uint, ctypes.POINTER(vp)]
    lib.crim_destroy.restype = None
    lib.crim_destroy.argtypes = [vp]
    lib.crim_upload.restype = ci
    lib.crim_upload.argtypes = [vp, vp, pd, vp, pd]
    lib.crim_upload_batch.restype = ci
    lib.crim_upload_batch.argtypes = [vp, ci, ci, vp, pd, pd, vp, pd, pd]
    lib.crim_upload_automask.restype = ci
    lib.crim_upload_automask.argtypes = [vp, ci, ci, vp, pd, pd, ci, ci, ci]
    lib.crim_upload_device.restype = ci
    lib.crim_upload_device.argtypes = [vp, ci, ci, vp, vp]
    lib.crim_split_virtual.restype = ci
    lib.crim_split_virtual.argtypes = [vp, ci]
    lib.crim_split_unique_id.restype = ci
    lib.crim_split_unique_id.argtypes = [vp]
    lib.crim_split_init.restype = ci
    lib.crim_split_init.argtypes = [vp, ci, ci, vp]
    lib.crim_split_init_host.restype = ci
    lib.crim_split_init_host.argtypes = [vp, ci, ci, ctypes.c_char_p]
    lib.crim_split_status.restype = ci
    lib.crim_split_status.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(cl)]
    lib.crim_split_info.restype = ci
    lib.crim_split_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.crim_run.restype = ci
    lib.crim_run.argtypes = [vp, cl, ctypes.POINTER(cl), ctypes.POINTER(cl)]
    lib.crim_download.restype = ci
    lib.crim_download.argtypes = [vp, vp, vp, vp]
    lib.crim_download_batch.restype = ci
    lib.crim_download_batch.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.crim_download_device.restype = ci
    lib.crim_download_device.argtypes = [vp, ci, ci, vp]
    lib.crim_trace.restype = ci
    lib.crim_trace.argtypes = [vp, ci, vp, cl, ctypes.POINTER(cl)]
    lib.crim_iterations.restype = ci
    lib.crim_iterations.argtypes = [vp, ci, ctypes.POINTER(cl)]
    lib.crim_search_times.restype = ci
    lib.crim_search_times.argtypes = [vp, ci, vp, cl, ctypes.POINTER(cl)]
    lib.crim_iteration_stamps.restype = ci
    lib.crim_iteration_stamps.argtypes = [vp, ci, vp, cl, ctypes.POINTER(cl)]
    lib.crim_noalloc_bytes_needed.restype = ctypes.c_size_t
    lib.crim_noalloc_bytes_needed.argtypes = [ci, ci, ci]
    lib.crim_debug_state.restype = ci
    lib.crim_debug_state.argtypes = [vp, ci, vp]
    lib.crim_debug_fill_tile_list.restype = ci
    lib.crim_debug_fill_tile_list.argtypes = [vp, ci]
    lib.crim_debug_dterm.restype = ci
    lib.crim_debug_dterm.argtypes = [vp, ci, vp]
    lib.crim_debug_stamps.restype = ci
    lib.crim_debug_stamps.argtypes = [vp, vp]
    lib.crim_lab_tables.restype = ci
    lib.crim_lab_tables.argtypes = [vp, vp, vp]
    lib.crim_profile.restype = ci
    lib.crim_profile.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(cl),
                                 ctypes.POINTER(ctypes.c_double)]
    lib.crim_host_alloc.restype = ci
    lib.crim_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.crim_host_free.restype = ci
    lib.crim_host_free.argtypes = [vp]
    lib.crim_inpaint_noalloc.restype = ci
    lib.crim_inpaint_noalloc.argtypes = [ci, ci, ci, ci, vp, vp, vp, ctypes.POINTER(cl)]
    lib.crim_filter_noise.restype = ci
    lib.crim_filter_noise.argtypes = [ci, ci, ci, ci, vp, vp, ci, ctypes.c_float, ci]
    lib.crim_filter_noise_device.restype = ci
    lib.crim_filter_noise_device.argtypes = [ci, ci, ci, ci, vp, vp, ci, ctypes.c_float, ci]
    lib.crim_filter_noise_engine.restype = ci
    lib.crim_filter_noise_engine.argtypes = [vp, ci, ctypes.c_float, ci]
    lib.crim_peek_fields.restype = ci
    lib.crim_peek_fields.argtypes = [vp, ci, vp, vp, vp]
    _lib = lib
    return lib


def _check(rc):
    if rc != CRIM_OK:
        msg = load_library().crim_last_error()
        raise CriminisiHipError(rc, msg.decode("utf-8", "replace") if msg else "")


def device_count():
    return int(load_library().crim_device_count())


def split_unique_id():
    """128-byte RCCL id (call on rank 0, share with the other ranks)."""
    buf = ctypes.create_string_buffer(128)
    _check(load_library().crim_split_unique_id(buf))
    return buf.raw


def lab_tables():
    """(gamma[256] u16, cbrt[3072] u16, coeffs[9] i32) the engine uses for BGR->Lab."""
    g = np.zeros(256, np.uint16)
    c = np.zeros(3072, np.uint16)
    k = np.zeros(9, np.int32)
    _check(load_library().crim_lab_tables(g.ctypes.data, c.ctypes.data, k.ctypes.data))
    return g, c, k


def pinned_empty(shape, dtype=np.uint8):
    """A NumPy array in page-locked host memory (uploads from / downloads into it are DMA
    transfers that overlap with kernels).  Release it with pinned_free(array) once no view of it
    is in use any more."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    ptr = ctypes.c_void_p()
    _check(load_library().crim_host_alloc(max(nbytes, 1), ctypes.byref(ptr)))
    buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(ptr.value)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def pinned_free(array):
    _check(load_library().crim_host_free(ctypes.c_void_p(array.ctypes.data)))


def filter_noise(images, median_ksize=3, nlm_h=7, apply_sharpening=True, device=0):
    """The reference's filter_noise (src/utils.py:106-138) on the GPU for one image [H,W,3] or a
    stack [n,H,W,3] of uint8 BGR images; returns the same shape."""
    lib = load_library()
    if lib.crim_device_count() <= 0:
        raise CriminisiHipError(CRIM_ERR_HIP, "no HIP device visible: the noise filter runs only on an AMD GPU")
    img = np.asarray(images)
    single = img.ndim == 3
    if single:
        img = img[None]
    if img.ndim != 4 or img.shape[3] != 3 or img.dtype != np.uint8:
        raise ValueError("filter_noise: expected uint8 images [n,H,W,3] in BGR order")
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    _check(lib.crim_filter_noise(int(device), img.shape[0], img.shape[1], img.shape[2], img.ctypes.data,
                                 out.ctypes.data, int(median_ksize), float(nlm_h), 1 if apply_sharpening else 0))
    return out[0] if single else out


NOALLOC_MAX_PIXELS = 512 * 512
NOALLOC_BYTES = 52 * NOALLOC_MAX_PIXELS + (1 << 18)


def noalloc_bytes_needed(height, width, patch_size):
    """Bytes of the static device array a crim_inpaint_noalloc call for this shape uses (no GPU needed)."""
    return int(load_library().crim_noalloc_bytes_needed(int(height), int(width), int(patch_size)))


def inpaint_noalloc(height, width, patch_size, d_bgr_ptr, d_mask_ptr, d_out_ptr, device=0):
    """One image, device buffers in and out, no allocation inside the library (state lives in a
    static device array; H*W <= NOALLOC_MAX_PIXELS).  Returns the iteration count."""
    it = ctypes.c_long()
    _check(load_library().crim_inpaint_noalloc(
        int(height), int(width), int(patch_size), int(device), ctypes.c_void_p(int(d_bgr_ptr)),
        ctypes.c_void_p(int(d_mask_ptr)), ctypes.c_void_p(int(d_out_ptr)), ctypes.byref(it)))
    return it.value


class Engine:
    """A batch of `batch` images of one shape resident on one GPU."""

    def __init__(self, height, width, patch_size=9, batch=1, device=0, flags=0):
        self._h = None
        lib = load_library()
        if lib.crim_device_count() <= 0:
            raise CriminisiHipError(CRIM_ERR_HIP, "no HIP device visible: the Criminisi engine "
                                    "runs only on an AMD GPU (there is no CPU fallback)")
        h = ctypes.c_void_p()
        _check(lib.crim_create_batch(int(batch), int(height), int(width), int(patch_size),
                                     int(device), int(flags), ctypes.byref(h)))
        self._h = h
        self.batch, self.height, self.width = int(batch), int(height), int(width)
        self.patch_size = int(patch_size)

    def close(self):
        if self._h is not None:
            load_library().crim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data in ----------------------------------------------------------------------------
    def upload(self, images, masks, first=0):
        """images: uint8 [n,H,W,3] (or [H,W,3]); masks: uint8 [n,H,W] (or [H,W]), nonzero = fill."""
        img = np.asarray(images)
        msk = np.asarray(masks)
        if img.ndim == 3:
            img = img[None]
        if msk.ndim == 2:
            msk = msk[None]
        img = np.ascontiguousarray(img, dtype=np.uint8)
        msk = np.ascontiguousarray(msk, dtype=np.uint8)
        n = img.shape[0]
        if img.shape != (n, self.height, self.width, 3) or msk.shape != (n, self.height, self.width):
            raise ValueError("upload: expected images [n,%d,%d,3] and masks [n,%d,%d], got %s and %s"
                             % (self.height, self.width, self.height, self.width, img.shape, msk.shape))
        _check(load_library().crim_upload_batch(
            self._h, int(first), n, img.ctypes.data, img.strides[0], img.strides[1],
            msk.ctypes.data, msk.strides[0], msk.strides[1]))

    def upload_automask(self, images, first=0, threshold=10, border_width=10, dilation_size=4):
        """Uploads images only; the mask is built on the GPU with the recipe of the reference's
        inpaint_image (src/utils.py:79-100)."""
        img = np.asarray(images)
        if img.ndim == 3:
            img = img[None]
        img = np.ascontiguousarray(img, dtype=np.uint8)
        n = img.shape[0]
        if img.shape != (n, self.height, self.width, 3):
            raise ValueError("upload_automask: expected images [n,%d,%d,3], got %s"
                             % (self.height, self.width, img.shape))
        _check(load_library().crim_upload_automask(
            self._h, int(first), n, img.ctypes.data, img.strides[0], img.strides[1],
            int(threshold), int(border_width), int(dilation_size)))

    def upload_device(self, d_bgr_ptr, d_mask_ptr, first=0, count=None):
        """Dense device buffers (raw pointers as ints) already on this engine's GPU."""
        count = self.batch - first if count is None else count
        _check(load_library().crim_upload_device(self._h, int(first), int(count),
                                                 ctypes.c_void_p(int(d_bgr_ptr)),
                                                 ctypes.c_void_p(int(d_mask_ptr))))

    # -- split search (one large image) -------------------------------------------------------
    def split_virtual(self, nparts):
        _check(load_library().crim_split_virtual(self._h, int(nparts)))

    def split_init(self, nranks, rank, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _check(load_library().crim_split_init(self._h, int(nranks), int(rank), buf))

    def split_init_host(self, nranks, rank, shared_file):
        """This process scores part `rank` of `nranks`; the ranks' processes (on one host) exchange their
        records through the file-backed mapping `shared_file` (a fresh path every rank passes)."""
        _check(load_library().crim_split_init_host(self._h, int(nranks), int(rank), os.fsencode(shared_file)))

    def split_status(self):
        """(iterations replayed from captured graphs?, records exchanged through host memory so far)."""
        g, n = ctypes.c_int(), ctypes.c_long()
        _check(load_library().crim_split_status(self._h, ctypes.byref(g), ctypes.byref(n)))
        return bool(g.value), n.value

    def split_info(self):
        """(ranks, rank) as the RCCL communicator reports them."""
        n, r = ctypes.c_int(), ctypes.c_int()
        _check(load_library().crim_split_info(self._h, ctypes.byref(n), ctypes.byref(r)))
        return n.value, r.value

    # -- loop -------------------------------------------------------------------------------
    def run(self, max_iters=0):
        it = ctypes.c_long()
        rem = ctypes.c_long()
        _check(load_library().crim_run(self._h, int(max_iters), ctypes.byref(it), ctypes.byref(rem)))
        return it.value, rem.value

    # -- data out ---------------------------------------------------------------------------
    def download(self, first=0, count=None, want_conf=False, want_mask=False, out=None):
        count = self.batch - first if count is None else count
        if out is None:
            out = np.empty((count, self.height, self.width, 3), np.uint8)
        elif out.shape != (count, self.height, self.width, 3) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("download: out must be a C-contiguous uint8 [%d,%d,%d,3] array"
                             % (count, self.height, self.width))
        conf = np.empty((count, self.height, self.width), np.float64) if want_conf else None
        msk = np.empty((count, self.height, self.width), np.uint8) if want_mask else None
        _check(load_library().crim_download_batch(
            self._h, int(first), int(count), out.ctypes.data,
            conf.ctypes.data if want_conf else None, msk.ctypes.data if want_mask else None))
        return out, conf, msk

    def download_device(self, d_out_ptr, first=0, count=None):
        count = self.batch - first if count is None else count
        _check(load_library().crim_download_device(self._h, int(first), int(count),
                                                   ctypes.c_void_p(int(d_out_ptr))))

    def filter_noise(self, median_ksize=3, nlm_h=7, apply_sharpening=True):
        """filter_noise (src/utils.py:106-138) in place on the engine's working images."""
        _check(load_library().crim_filter_noise_engine(self._h, int(median_ksize), float(nlm_h),
                                                       1 if apply_sharpening else 0))

    def iterations(self, index=0):
        n = ctypes.c_long()
        _check(load_library().crim_iterations(self._h, int(index), ctypes.byref(n)))
        return n.value

    def trace(self, index=0):
        n = ctypes.c_long()
        _check(load_library().crim_trace(self._h, int(index), None, 0, ctypes.byref(n)))
        buf = np.zeros(max(n.value, 1), ITER_REC_DTYPE)
        _check(load_library().crim_trace(self._h, int(index), buf.ctypes.data, n.value, ctypes.byref(n)))
        return buf[:n.value]

    def search_times(self, index=0):
        """Microseconds each executed iteration spent between the end of its target selection and the start
        of its update (the search launch), as float32; empty under FLAG_NO_TRACE."""
        n = ctypes.c_long()
        _check(load_library().crim_search_times(self._h, int(index), None, 0, ctypes.byref(n)))
        buf = np.zeros(max(n.value, 1), np.float32)
        _check(load_library().crim_search_times(self._h, int(index), buf.ctypes.data, n.value, ctypes.byref(n)))
        return buf[:n.value]

    def iteration_stamps(self, index=0):
        """int64 [n,3] per executed iteration: 100 MHz device clock at the start of the selecting launch, at
        the end of the selection, at the start of the update."""
        n = ctypes.c_long()
        _check(load_library().crim_iteration_stamps(self._h, int(index), None, 0, ctypes.byref(n)))
        buf = np.zeros((max(n.value, 1), 3), np.int64)
        _check(load_library().crim_iteration_stamps(self._h, int(index), buf.ctypes.data, n.value, ctypes.byref(n)))
        return buf[:n.value]

    def profile(self):
        ms = ctypes.c_double()
        launches = ctypes.c_long()
        evals = ctypes.c_double()
        _check(load_library().crim_profile(self._h, ctypes.byref(ms), ctypes.byref(launches),
                                           ctypes.byref(evals)))
        return ms.value, launches.value, evals.value

    def peek_fields(self, index=0):
        """front (u8), data (f64), priority (f64) the next iteration would compute."""
        front = np.empty((self.height, self.width), np.uint8)
        data = np.empty((self.height, self.width), np.float64)
        prio = np.empty((self.height, self.width), np.float64)
        _check(load_library().crim_peek_fields(self._h, int(index), front.ctypes.data,
                                               data.ctypes.data, prio.ctypes.data))
        return front, data, prio
