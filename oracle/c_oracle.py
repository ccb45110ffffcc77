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
        lib.crim_oracle_search.restype = ctypes.c_int
        lib.crim_oracle_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def search(image_bgr, mask, patch_size, tr, tc):
    """The search of one iteration alone (criminisi_inpainter.py:220-265) for the given state and
    target pixel: (sy, sx, cost)."""
    img = np.ascontiguousarray(image_bgr, dtype=np.uint8)
    msk = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    H, W = msk.shape
    gam = np.ascontiguousarray(cvr.SRGB_GAMMA_TAB, dtype=np.uint16)
    cb = np.ascontiguousarray(cvr.LAB_CBRT_TAB, dtype=np.uint16)
    co = np.ascontiguousarray(cvr.LAB_COEFFS, dtype=np.int32)
    sy, sx, cost = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
    rc = _lib().crim_oracle_search(img.ctypes.data, msk.ctypes.data, H, W, int(patch_size), int(tr), int(tc),
                                   gam.ctypes.data, cb.ctypes.data, co.ctypes.data,
                                   ctypes.byref(sy), ctypes.byref(sx), ctypes.byref(cost))
    if rc:
        raise MemoryError("crim_oracle_search: allocation failed")
    return sy.value, sx.value, cost.value


def replay(image_bgr, mask, patch_size, trace, upto):
    """Image and mask after the first `upto` iterations of a run, rebuilt from its trace alone
    (criminisi_inpainter.py:267-308: masked pixels of the target patch take the source patch's
    pixels, then the patch's mask is cleared)."""
    img = np.array(image_bgr, dtype=np.uint8, copy=True)
    msk = np.array(np.asarray(mask) != 0, dtype=np.uint8)
    H, W = msk.shape
    half = (int(patch_size) - 1) // 2
    for k in range(int(upto)):
        tr, tc, sy, sx = (int(trace[f][k]) for f in ("tr", "tc", "sy", "sx"))
        r0, r1 = max(0, tr - half), min(tr + half, H - 1)
        c0, c1 = max(0, tc - half), min(tc + half, W - 1)
        ph, pw = r1 - r0 + 1, c1 - c0 + 1
        src = img[sy:sy + ph, sx:sx + pw].copy()
        sel = msk[r0:r1 + 1, c0:c1 + 1] != 0
        img[r0:r1 + 1, c0:c1 + 1][sel] = src[sel]
        msk[r0:r1 + 1, c0:c1 + 1] = 0
    return img, msk


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
