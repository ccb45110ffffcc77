"""ctypes wrapper around oracle/libcrim_oracle.so (the plain-C restatement of the reference loop).

TEST INFRASTRUCTURE ONLY -- see oracle/cv2_restated.py for the rules.  Parity unpinned.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import cv2_restated as cvr

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REC_DTYPE = np.dtype([("tr", "<i4"), ("tc", "<i4"), ("sy", "<i4"), ("sx", "<i4"),
                      ("cost", "<f8"), ("remaining", "<i8")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcrim_oracle.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcrim_oracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.crim_oracle_run.restype = ctypes.c_long
        lib.crim_oracle_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_long]
        _LIB = lib
    return _LIB


def run(image_bgr, mask, patch_size=9, max_iters=0, want_trace=True):
    """Returns (out_bgr, out_conf, out_mask, trace_records, n_iters)."""
    img = np.ascontiguousarray(image_bgr, dtype=np.uint8)
    msk = np.ascontiguousarray(np.asarray(mask).round() != 0, dtype=np.uint8)
    H, W = msk.shape
    assert img.shape == (H, W, 3)
    gam = np.ascontiguousarray(cvr.SRGB_GAMMA_TAB, dtype=np.uint16)
    cb = np.ascontiguousarray(cvr.LAB_CBRT_TAB, dtype=np.uint16)
    co = np.ascontiguousarray(cvr.LAB_COEFFS, dtype=np.int32)
    out = np.empty_like(img)
    conf = np.empty((H, W), dtype=np.float64)
    omask = np.empty((H, W), dtype=np.uint8)
    cap = int(msk.sum()) + 1 if want_trace else 0
    if max_iters and max_iters > 0:
        cap = min(cap, int(max_iters))
    trace = np.zeros(max(cap, 1), dtype=REC_DTYPE)
    n = _lib().crim_oracle_run(img.ctypes.data, msk.ctypes.data, H, W, int(patch_size),
                               gam.ctypes.data, cb.ctypes.data, co.ctypes.data, int(max_iters),
                               out.ctypes.data, conf.ctypes.data, omask.ctypes.data,
                               trace.ctypes.data if want_trace else None, cap)
    if n < 0:
        raise MemoryError("crim_oracle_run: allocation failed")
    return out, conf, omask, trace[:min(n, cap)], int(n)
