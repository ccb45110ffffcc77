"""NumPy restatement of the reference's noise-filter stage, ``filter_noise`` in
/root/reference/src/utils.py:106-138: ``cv2.medianBlur`` (:118), ``cv2.fastNlMeansDenoisingColored
(image, None, h=nlm_h, hColor=1, templateWindowSize=9, searchWindowSize=31)`` (:124-126), and the
Laplacian sharpening ``cv2.filter2D(image, cv2.CV_8U, lap_kernel)`` + ``cv2.subtract`` (:128-136).

TEST INFRASTRUCTURE ONLY (see oracle/cv2_restated.py for the rules).

PARITY UNPINNED.  All arithmetic of this stage lives in OpenCV (opencv-python >= 4.13, not
installed here, no fixtures in the reference).  Everything below restates OpenCV's published
algorithms from memory of its sources (modules/photo/src/denoising.cpp,
fast_nlmeans_denoising_invoker.hpp, fast_nlmeans_denoising_invoker_commons.hpp,
modules/imgproc/src/color_lab.cpp, median_blur, filter2D); every constant marked [unpinned] is
recalled, not checked.  tests/test_cv2_pins.py compares each piece with the real library wherever
OpenCV can be imported.

What is restated:
 * medianBlur: per-channel k x k median, BORDER_REPLICATE.
 * fastNlMeansDenoisingColored: BGR -> Lab with COLOR_LBGR2Lab (the LINEAR variant: no sRGB gamma)
   [unpinned], non-local means on L with h, on the 2-channel (a, b) image with hColor, then
   COLOR_Lab2LBGR.
 * the 8-bit non-local means itself: the image is extended by search/2 + template/2 pixels with
   BORDER_REFLECT_101; the distance between two template windows is the sum of squared differences
   over the window (and over the channels); it is divided by the smallest power of two >= template^2
   (a shift) instead of template^2 and looked up in a table of integer weights
   round(fixed_point_mult * exp(-dist / (h*h*channels))) with weights below 0.001*fixed_point_mult
   set to 0 and fixed_point_mult = INT_MAX / (search^2 * 255); the result is the weighted mean with
   rounding (sum + wsum/2) / wsum in unsigned arithmetic.
 * sharpening: correlation with [[0,1,0],[1,-4,1],[0,1,0]], BORDER_REFLECT_101, saturated to uint8
   (negative responses become 0), then a saturating subtraction.
"""
import math

import numpy as np

from . import cv2_restated as cvr

INT_MAX = 2 ** 31 - 1


# ---------------------------------------------------------------------------------------------
# median
# ---------------------------------------------------------------------------------------------

def median_blur(bgr, ksize=3):
    """cv2.medianBlur(image, ksize) for uint8 -- utils.py:118."""
    if ksize == 1:
        return bgr.copy()
    r = ksize // 2
    pad = np.pad(bgr, ((r, r), (r, r), (0, 0)), mode="edge")
    h, w = bgr.shape[:2]
    stack = np.stack([pad[i:i + h, j:j + w] for i in range(ksize) for j in range(ksize)], axis=0)
    stack.sort(axis=0)
    return np.ascontiguousarray(stack[(ksize * ksize) // 2])


# ---------------------------------------------------------------------------------------------
# linear BGR <-> Lab, 8 bit
# ---------------------------------------------------------------------------------------------

def lbgr2lab_u8(bgr):
    """cv2.cvtColor(img, cv2.COLOR_LBGR2Lab) for uint8: the integer pipeline of
    cv2_restated.bgr2lab_u8 with the linear table i * 8 in place of the sRGB gamma table
    [unpinned]."""
    c = cvr.LAB_COEFFS
    lin = np.arange(256, dtype=np.int64) * (1 << cvr.GAMMA_SHIFT)
    B = lin[bgr[:, :, 0]]
    G = lin[bgr[:, :, 1]]
    R = lin[bgr[:, :, 2]]
    half = 1 << (cvr.LAB_SHIFT - 1)
    fX = cvr.LAB_CBRT_TAB[(B * c[0] + G * c[1] + R * c[2] + half) >> cvr.LAB_SHIFT].astype(np.int64)
    fY = cvr.LAB_CBRT_TAB[(B * c[3] + G * c[4] + R * c[5] + half) >> cvr.LAB_SHIFT].astype(np.int64)
    fZ = cvr.LAB_CBRT_TAB[(B * c[6] + G * c[7] + R * c[8] + half) >> cvr.LAB_SHIFT].astype(np.int64)
    half2 = 1 << (cvr.LAB_SHIFT2 - 1)
    L = (296 * fY - 1336934 + half2) >> cvr.LAB_SHIFT2
    a = (500 * (fX - fY) + 128 * (1 << cvr.LAB_SHIFT2) + half2) >> cvr.LAB_SHIFT2
    b = (200 * (fY - fZ) + 128 * (1 << cvr.LAB_SHIFT2) + half2) >> cvr.LAB_SHIFT2
    return np.clip(np.stack([L, a, b], axis=2), 0, 255).astype(np.uint8)


LAB_BASE_SHIFT = 14
LAB_BASE = 1 << LAB_BASE_SHIFT
INV_GAMMA_SHIFT = 12
MIN_AB_VALUE = -8145                 # [unpinned]
_XYZ2SRGB_D65 = (3.240479, -1.53715, -0.498535,
                 -0.969256, 1.875991, 0.041556,
                 0.055648, -0.204043, 1.057311)


def _cv_round_f32(x):
    return int(np.rint(np.float64(np.float32(x))))


def _build_lab_to_yf():
    """LabToYF_b: for every 8-bit L the pair (y, f(y)) in units of 1/16384, built with float32
    arithmetic [unpinned]."""
    f = np.float32
    tab = np.zeros((256, 2), np.int64)
    base = LAB_BASE
    for i in range(256):
        if i <= 20:
            y = _cv_round_f32(f(i * base * 20 * 9) / f(17 * 29 * 29 * 29))
            ify = _cv_round_f32(f(base) * (f(16) / f(116) + f(i * 5) / f(3 * 17 * 29)))
        else:
            fy = f(i * 100 * base) / f(255 * 116) + f(16 * base) / f(116)
            ify = _cv_round_f32(fy)
            y = _cv_round_f32(fy * fy * fy / f(base * base))
        tab[i, 0] = y
        tab[i, 1] = ify
    return tab


LAB_TO_YF = _build_lab_to_yf()


def _trunc_div(a, b):
    """C integer division (truncation toward zero) on int64 arrays, b > 0."""
    return np.where(a >= 0, a // b, -((-a) // b))


def ab_to_xz(v):
    """abToXZ_b[v - minABvalue] computed directly: the inverse of f() in units of 1/16384 with C
    integer arithmetic [unpinned]."""
    v = np.asarray(v, dtype=np.int64)
    lin = _trunc_div(v * 108, 841) - (((LAB_BASE * 16) // 116) * 108) // 841
    cube = _trunc_div(_trunc_div(v * v, LAB_BASE) * v, LAB_BASE)
    return np.where(v <= 3390, lin, cube)


def _lab2rgb_coeffs():
    wp = cvr._D65
    c = np.zeros(9, np.int64)
    for i in range(3):          # output channel i of B,G,R uses the XYZ->sRGB row 2 - i
        row = _XYZ2SRGB_D65[3 * (2 - i):3 * (2 - i) + 3]
        for j in range(3):
            c[3 * i + j] = int(np.rint(4096.0 * row[j] * wp[j]))
    return c


LAB2RGB_COEFFS = _lab2rgb_coeffs()


def lab2lbgr_u8(lab):
    """cv2.cvtColor(img, cv2.COLOR_Lab2LBGR) for uint8: OpenCV's integer Lab -> RGB path without
    the inverse sRGB gamma [unpinned]."""
    L = lab[:, :, 0].astype(np.int64)
    a = lab[:, :, 1].astype(np.int64)
    b = lab[:, :, 2].astype(np.int64)
    y = LAB_TO_YF[L, 0]
    ify = LAB_TO_YF[L, 1]
    adiv = ((5 * a * 53687 + (1 << 7)) >> 13) - (128 * LAB_BASE) // 500
    bdiv = ((b * 41943 + (1 << 4)) >> 9) - (128 * LAB_BASE) // 200 - 1
    x = ab_to_xz(ify + adiv)
    z = ab_to_xz(ify - bdiv)
    c = LAB2RGB_COEFFS
    shift = cvr.LAB_SHIFT + (LAB_BASE_SHIFT - INV_GAMMA_SHIFT)
    out = []
    for i in range(3):
        v = (x * c[3 * i] + y * c[3 * i + 1] + z * c[3 * i + 2] + (1 << (shift - 1))) >> shift
        v = np.clip(v, 0, (1 << INV_GAMMA_SHIFT) - 1)
        out.append((v * 255) >> INV_GAMMA_SHIFT)
    return np.stack(out, axis=2).astype(np.uint8)


# ---------------------------------------------------------------------------------------------
# non-local means, 8 bit
# ---------------------------------------------------------------------------------------------

def nlm_shift(template_window_size):
    p = 0
    while (1 << p) < template_window_size * template_window_size:
        p += 1
    return p


def nlm_fixed_point_mult(search_window_size):
    return min(INT_MAX // (search_window_size * search_window_size * 255), INT_MAX)


def nlm_weight_table(h, channels, template_window_size=9, search_window_size=31):
    """almost_dist2weight_: integer weight for every possible shifted distance."""
    tsq = template_window_size * template_window_size
    shift = nlm_shift(template_window_size)
    mult = float(1 << shift) / tsq
    max_dist = 255 * 255 * channels
    n = int(max_dist / mult + 1)
    fpm = nlm_fixed_point_mult(search_window_size)
    hh = float(np.float32(h) * np.float32(h) * np.float32(channels))
    tab = np.zeros(n, np.int64)
    for a in range(n):
        dist = a * mult
        w = math.exp(-dist / hh) if hh > 0 else 1.0     # libm's exp, as the library under test calls it
        weight = int(np.rint(fpm * w))
        if weight < 0.001 * fpm:
            weight = 0
        tab[a] = weight
    return tab


def _reflect101_pad(img, r):
    """copyMakeBorder(..., BORDER_REFLECT_101) by r pixels, also when r exceeds the image size."""
    h, w = img.shape[:2]

    def idx(n):
        i = np.arange(-r, n + r)
        if n == 1:
            return np.zeros_like(i)
        period = 2 * (n - 1)
        i = np.mod(i, period)
        return np.where(i >= n, period - i, i)

    return img[idx(h)][:, idx(w)]


def fast_nl_means_u8(img, h, template_window_size=9, search_window_size=31):
    """cv2.fastNlMeansDenoising for a uint8 image of 1 or 2 channels ([H,W] or [H,W,C])."""
    src = img if img.ndim == 3 else img[:, :, None]
    H, W, C = src.shape
    twh, swh = template_window_size // 2, search_window_size // 2
    border = twh + swh
    ext = _reflect101_pad(src, border).astype(np.int64)
    shift = nlm_shift(template_window_size)
    tab = nlm_weight_table(h, C, template_window_size, search_window_size)
    est = np.zeros((H, W, C), np.int64)
    wsum = np.zeros((H, W), np.int64)
    tw = template_window_size
    ref = ext[border - twh:border + H + twh, border - twh:border + W + twh]
    for dy in range(-swh, swh + 1):
        for dx in range(-swh, swh + 1):
            cand = ext[border + dy - twh:border + dy + H + twh, border + dx - twh:border + dx + W + twh]
            d = ((ref - cand) ** 2).sum(axis=2)
            s = np.zeros((d.shape[0] + 1, d.shape[1] + 1), np.int64)
            s[1:, 1:] = d.cumsum(0).cumsum(1)
            dist = s[tw:tw + H, tw:tw + W] - s[0:H, tw:tw + W] - s[tw:tw + H, 0:W] + s[0:H, 0:W]
            wgt = tab[dist >> shift]
            est += wgt[:, :, None] * ext[border + dy:border + dy + H, border + dx:border + dx + W]
            wsum += wgt
    out = (est + (wsum // 2)[:, :, None]) // wsum[:, :, None]
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


def fast_nl_means_colored(bgr, h, h_color, template_window_size=9, search_window_size=31):
    """cv2.fastNlMeansDenoisingColored -- utils.py:124-126."""
    lab = lbgr2lab_u8(bgr)
    l = fast_nl_means_u8(lab[:, :, 0], h, template_window_size, search_window_size)
    ab = fast_nl_means_u8(lab[:, :, 1:3], h_color, template_window_size, search_window_size)
    return lab2lbgr_u8(np.concatenate([l[:, :, None], ab], axis=2))


# ---------------------------------------------------------------------------------------------
# sharpening
# ---------------------------------------------------------------------------------------------

def laplacian_sharpen(bgr):
    """image - sat_u8(filter2D(image, CV_8U, [[0,1,0],[1,-4,1],[0,1,0]])), saturating --
    utils.py:128-136."""
    p = _reflect101_pad(bgr, 1).astype(np.int64)
    h, w = bgr.shape[:2]
    lap = p[0:h, 1:w + 1] + p[2:h + 2, 1:w + 1] + p[1:h + 1, 0:w] + p[1:h + 1, 2:w + 2] - 4 * p[1:h + 1, 1:w + 1]
    lap = np.clip(lap, 0, 255)
    return np.clip(bgr.astype(np.int64) - lap, 0, 255).astype(np.uint8)


def filter_noise(image, median_ksize=3, nlm_h=7, apply_sharpening=True):
    """filter_noise -- utils.py:106-138."""
    image = median_blur(image, median_ksize)
    image = fast_nl_means_colored(image, nlm_h, 1, 9, 31)
    if apply_sharpening:
        image = laplacian_sharpen(image)
    return image
