"""Shared helpers for the test-suite."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-processing_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pipeline_mask(bgr, dilation_size=4, border_width=10):
    """NumPy restatement of the mask recipe in /root/reference/src/utils.py:79-100:
    grey <= 10 -> zero a 10-pixel frame -> dilate with a k x k box (OpenCV anchors an even box
    at k//2, so offsets run from -k//2 to k - k//2 - 1) -> {0,1}."""
    from oracle import cv2_restated as cvr
    grey = cvr.bgr2gray_u8(bgr)
    m = (grey <= 10).astype(np.uint8)
    m[:border_width, :] = 0
    m[-border_width:, :] = 0
    m[:, :border_width] = 0
    m[:, -border_width:] = 0
    k = dilation_size
    a = k // 2
    h, w = m.shape
    pad = np.zeros((h + k, w + k), np.uint8)
    pad[a:a + h, a:a + w] = m
    out = np.zeros_like(m)
    for dy in range(k):
        for dx in range(k):
            out |= pad[dy:dy + h, dx:dx + w]
    return out


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


class HipBuffers:
    """Plain device allocations for tests, through the HIP runtime the engine library itself is
    linked against (ctypes; no PyTorch, whose bundled runtime cannot initialise a GPU that the
    system runtime already drives in the same process)."""

    def __init__(self):
        import ctypes
        self._ct = ctypes
        self._hip = ctypes.CDLL("libamdhip64.so.7")
        self._hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self._hip.hipFree.argtypes = [ctypes.c_void_p]
        self._hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self._hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        self._live = []

    def alloc(self, nbytes):
        ptr = self._ct.c_void_p()
        rc = self._hip.hipMalloc(self._ct.byref(ptr), max(int(nbytes), 1))
        assert rc == 0 and ptr.value, "hipMalloc failed (%d)" % rc
        self._live.append(ptr.value)
        return ptr.value

    def to_device(self, array):
        a = np.ascontiguousarray(array)
        ptr = self.alloc(a.nbytes)
        assert self._hip.hipMemcpy(ptr, a.ctypes.data, a.nbytes, 1) == 0     # host to device
        return ptr

    def zeros(self, nbytes):
        ptr = self.alloc(nbytes)
        assert self._hip.hipMemset(ptr, 0, int(nbytes)) == 0
        return ptr

    def to_host(self, ptr, shape, dtype=np.uint8):
        out = np.empty(shape, dtype)
        assert self._hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0  # device to host
        return out

    def free_all(self):
        for ptr in self._live:
            self._hip.hipFree(ptr)
        self._live = []
