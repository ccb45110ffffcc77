"""CPU restatements of the third-party primitives the reference's inpainting path calls.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.

PARITY UNPINNED: the reference (``/root/reference/src/criminisi_inpainter.py``) does its
arithmetic inside ``opencv-python >= 4.13`` (``pyproject.toml:6-11``, no lock file), which is not
installed in the build container, and the reference holds no tests, fixtures or golden vectors
for this path.  Every function below restates the published OpenCV algorithm from its
documented behaviour; where that behaviour could be checked against something present here
(NumPy itself, hand-derivable answers) the tests do so.  Each function names the reference call
site it stands in for.
"""
from fractions import Fraction

import numpy as np

# ----------------------------------------------------------------------------------------------
# Border handling shared by cv2.Laplacian / cv2.filter2D defaults (BORDER_REFLECT_101).
# ----------------------------------------------------------------------------------------------


def _pad_reflect101(a, r):
    """gfedcb|abcdefgh|gfedcba padding by r pixels (the edge pixel is not repeated)."""
    return np.pad(a, r, mode="reflect")


def laplacian_ksize3_u8(mask_u8):
    """cv2.Laplacian(mask, cv2.CV_8U, ksize=3) -- criminisi_inpainter.py:103.

    OpenCV's ksize=3 Laplacian aperture is [[2,0,2],[0,-8,0],[2,0,2]] (not the 4-neighbour
    cross, which is ksize=1), default border BORDER_REFLECT_101, result saturated to uint8.
    """
    m = _pad_reflect101(mask_u8.astype(np.int32), 1)
    h, w = mask_u8.shape
    acc = 2 * (m[0:h, 0:w] + m[0:h, 2:w + 2] + m[2:h + 2, 0:w] + m[2:h + 2, 2:w + 2])
    acc = acc - 8 * m[1:h + 1, 1:w + 1]
    return np.clip(acc, 0, 255).astype(np.uint8)


def filter2d_3x3_f64(src_f64, kernel):
    """cv2.filter2D(src.astype(float), -1, 3x3 kernel) -- criminisi_inpainter.py:149-150.

    Correlation (kernel not flipped), anchor at the kernel centre, BORDER_REFLECT_101, float64
    in and out.  The accumulation order does not matter here: the call sites feed a 0/1 mask
    and kernels whose entries are multiples of 1/4, so every partial sum is exact.
    """
    s = _pad_reflect101(np.asarray(src_f64, dtype=np.float64), 1)
    h, w = src_f64.shape
    out = np.zeros((h, w), dtype=np.float64)
    for i in range(3):
        for j in range(3):
            if kernel[i][j] != 0:
                out += kernel[i][j] * s[i:i + h, j:j + w]
    return out


def bgr2gray_u8(bgr):
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) for uint8 -- criminisi_inpainter.py:170.

    OpenCV 4.x fixed point: 15-bit coefficients B 3735, G 19235, R 9798, round to nearest.
    """
    b = bgr[:, :, 0].astype(np.int64)
    g = bgr[:, :, 1].astype(np.int64)
    r = bgr[:, :, 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


# ----------------------------------------------------------------------------------------------
# BGR -> CIE L*a*b* for uint8 (OpenCV's integer "RGB2Lab_b" path, sRGB gamma on).
# ----------------------------------------------------------------------------------------------

GAMMA_SHIFT = 3
LAB_SHIFT = 12
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
CBRT_TAB_SIZE = 256 * 3 // 2 * (1 << GAMMA_SHIFT)  # 3072

_SRGB2XYZ_D65 = (0.412453, 0.357580, 0.180423,
                 0.212671, 0.715160, 0.072169,
                 0.019334, 0.119193, 0.950227)
_D65 = (0.950456, 1.0, 1.088754)


def _round_half_even(x):
    """cvRound semantics (round to nearest, ties to even) on a Python float / numpy scalar."""
    return int(np.rint(np.float64(x)))


def _fraction_to_f32(v):
    """Correctly rounded float32 of an exact rational (avoids double rounding through f64)."""
    f = np.float32(float(v))
    best = f
    for cand in (np.nextafter(f, np.float32(-np.inf)), np.nextafter(f, np.float32(np.inf))):
        d_best = abs(Fraction(float(best)) - v)
        d_cand = abs(Fraction(float(cand)) - v)
        if d_cand < d_best:
            best = cand
        elif d_cand == d_best:
            # tie: even mantissa wins
            if (np.float32(cand).view(np.uint32) & 1) == 0:
                best = cand
    return np.float32(best)


def opencv_cbrt_f32(x):
    """OpenCV's cubeRoot(): exponent split + quartic rational polynomial evaluated in double,
    rounded to float32.  Not a correctly rounded cbrt, which is why it is restated rather than
    replaced by np.cbrt: the Lab table entries are cvRound(32768 * this)."""
    x = np.float32(x)
    if x == 0:
        return np.float32(0)
    bits = int(np.float32(x).view(np.uint32))
    ix = bits & 0x7FFFFFFF
    ex = (ix >> 23) - 127
    shx = int(np.fmod(ex, 3))          # C remainder: sign follows the dividend
    shx -= 3 if shx >= 0 else 0
    ex = (ex - shx) // 3               # exact division
    frbits = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23)
    fr = float(np.uint32(frbits).view(np.float32))   # 0.125 <= fr < 1
    num = ((((45.2548339756803022511987494 * fr +
              192.2798368355061050458134625) * fr +
             119.1654824285581628956914143) * fr +
            13.43250139086239872172837314) * fr +
           0.1636161226585754240958355063)
    den = ((((14.80884093219134573786480845 * fr +
              151.9714051044435648658557668) * fr +
             168.5254414101568283957668343) * fr +
            33.9905941350215598754191872) * fr +
           1.0)
    y = np.float32(num / den)
    ybits = int(y.view(np.uint32)) + (ex << 23)
    return np.uint32(ybits).view(np.float32)


def _build_lab_tables():
    f255 = np.float32(255)
    int_scale = np.float32(255 * (1 << GAMMA_SHIFT))          # 2040
    gamma_tab = np.zeros(256, dtype=np.uint16)
    one_plus_shift = 1.0 + 11.0 / 200.0
    for i in range(256):
        x = np.float32(i) / f255
        xd = float(x)
        if xd <= 809.0 / 20000.0:
            g = xd / (323.0 / 25.0)
        else:
            g = float(np.power(np.float64((xd + 11.0 / 200.0) / one_plus_shift),
                               np.float64(12.0 / 5.0)))
        gamma_tab[i] = _round_half_even(int_scale * np.float32(g))

    lthresh = np.float32(216) / np.float32(24389)
    lscale = np.float32(841) / np.float32(108)
    lbias = np.float32(16) / np.float32(116)
    cb_scale = np.float32(1) / (f255 * np.float32(1 << GAMMA_SHIFT))
    lshift2 = np.float32(1 << LAB_SHIFT2)
    cbrt_tab = np.zeros(CBRT_TAB_SIZE, dtype=np.uint16)
    for i in range(CBRT_TAB_SIZE):
        x = cb_scale * np.float32(i)
        if x < lthresh:
            # fused multiply-add in float32
            v = Fraction(float(x)) * Fraction(float(lscale)) + Fraction(float(lbias))
            y = _fraction_to_f32(v)
        else:
            y = opencv_cbrt_f32(x)
        cbrt_tab[i] = _round_half_even(lshift2 * y)

    coeffs = np.zeros(9, dtype=np.int64)   # laid out for B,G,R source order
    for i in range(3):
        c = _SRGB2XYZ_D65[3 * i:3 * i + 3]
        coeffs[3 * i + 2] = _round_half_even(4096.0 * c[0] / _D65[i])   # multiplies R
        coeffs[3 * i + 1] = _round_half_even(4096.0 * c[1] / _D65[i])   # multiplies G
        coeffs[3 * i + 0] = _round_half_even(4096.0 * c[2] / _D65[i])   # multiplies B
    return gamma_tab, cbrt_tab, coeffs


SRGB_GAMMA_TAB, LAB_CBRT_TAB, LAB_COEFFS = _build_lab_tables()


def bgr2lab_u8(bgr):
    """cv2.cvtColor(img, cv2.COLOR_BGR2LAB) for uint8 -- criminisi_inpainter.py:225.

    Integer pipeline: sRGB gamma LUT (x8 scale) -> 12-bit XYZ matrix normalised by the D65
    white point -> cube-root LUT (15-bit) -> L = 296*fY - 1336934, a = 500*(fX-fY)+128,
    b = 200*(fY-fZ)+128, each descaled by 15 bits with rounding and saturated to uint8.
    """
    c = LAB_COEFFS
    B = SRGB_GAMMA_TAB[bgr[:, :, 0]].astype(np.int64)
    G = SRGB_GAMMA_TAB[bgr[:, :, 1]].astype(np.int64)
    R = SRGB_GAMMA_TAB[bgr[:, :, 2]].astype(np.int64)
    half = 1 << (LAB_SHIFT - 1)
    fX = LAB_CBRT_TAB[(B * c[0] + G * c[1] + R * c[2] + half) >> LAB_SHIFT].astype(np.int64)
    fY = LAB_CBRT_TAB[(B * c[3] + G * c[4] + R * c[5] + half) >> LAB_SHIFT].astype(np.int64)
    fZ = LAB_CBRT_TAB[(B * c[6] + G * c[7] + R * c[8] + half) >> LAB_SHIFT].astype(np.int64)
    lscale = (116 * 255 + 50) // 100
    lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)
    half2 = 1 << (LAB_SHIFT2 - 1)
    L = (lscale * fY + lshift + half2) >> LAB_SHIFT2
    a = (500 * (fX - fY) + 128 * (1 << LAB_SHIFT2) + half2) >> LAB_SHIFT2
    b = (200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2) + half2) >> LAB_SHIFT2
    out = np.stack([L, a, b], axis=2)
    return np.clip(out, 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------------------------
# Template correlation and window sums.
# ----------------------------------------------------------------------------------------------


def match_template_ccorr_exact(image_i64, templ_i64):
    """cv2.matchTemplate(I, T, cv2.TM_CCORR) -- criminisi_inpainter.py:245-246.

    R[y, x] = sum_{i,j} T[i, j] * I[y+i, x+j] over the valid region only.  The reference feeds
    float32 planes holding integers; every true result at its call sites is an integer below
    2**24 for patch_size <= 16, so the exact integer correlation is the canonical value.
    """
    th, tw = templ_i64.shape
    h, w = image_i64.shape
    oh, ow = h - th + 1, w - tw + 1
    out = np.zeros((oh, ow), dtype=np.int64)
    for i in range(th):
        for j in range(tw):
            t = int(templ_i64[i, j])
            if t != 0:
                out += t * image_i64[i:i + oh, j:j + ow]
    return out


def window_sum_topleft(mask_u8, ph, pw):
    """cv2.filter2D(mask.astype(float32), -1, ones((ph,pw)), anchor=(0,0),
    borderType=BORDER_CONSTANT) -- criminisi_inpainter.py:250-254.

    out[y, x] = sum of mask[y:y+ph, x:x+pw] with zeros beyond the image; same size as mask.
    """
    h, w = mask_u8.shape
    m = np.zeros((h + ph, w + pw), dtype=np.int64)
    m[:h, :w] = mask_u8
    sat = np.zeros((h + ph + 1, w + pw + 1), dtype=np.int64)
    sat[1:, 1:] = m.cumsum(0).cumsum(1)
    return (sat[ph:ph + h, pw:pw + w] - sat[0:h, pw:pw + w]
            - sat[ph:ph + h, 0:w] + sat[0:h, 0:w])
