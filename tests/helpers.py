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
