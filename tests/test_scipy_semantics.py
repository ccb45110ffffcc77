"""The generic semantics of the restated primitives (correlation without kernel flip, mirror
border that does not repeat the edge pixel, valid-region template correlation, box dilation
window) against SciPy's independent implementations of the same textbook operations.  This does
not pin OpenCV-specific constants (aperture of Laplacian ksize 3, BGR2GRAY / BGR2LAB tables,
anchor of an even dilation box): tests/test_cv2_pins.py does that wherever OpenCV exists."""
import numpy as np
import pytest
from scipy import ndimage, signal

from helpers import pipeline_mask
from oracle import cv2_restated as cvr


def _masks():
    rng = np.random.default_rng(15)
    for h, w in ((5, 5), (16, 23), (40, 37), (2, 9), (9, 2)):
        yield (rng.random((h, w)) < 0.3).astype(np.uint8)


def test_3x3_correlation_with_mirror_border():
    xk = np.array([[.25, 0, -.25], [.5, 0, -.5], [.25, 0, -.25]])
    yk = np.array([[-.25, -.5, -.25], [0, 0, 0], [.25, .5, .25]])
    for m in _masks():
        for k in (xk, yk):
            want = ndimage.correlate(m.astype(np.float64), k, mode="mirror")
            assert np.array_equal(cvr.filter2d_3x3_f64(m.astype(np.float64), k), want)


def test_laplacian_aperture_as_a_mirror_correlation():
    k = np.array([[2, 0, 2], [0, -8, 0], [2, 0, 2]], np.int64)
    for m in _masks():
        want = np.clip(ndimage.correlate(m.astype(np.int64), k, mode="mirror"), 0, 255).astype(np.uint8)
        assert np.array_equal(cvr.laplacian_ksize3_u8(m), want)


@pytest.mark.parametrize("p", [3, 9, 15])
def test_template_correlation_over_the_valid_region(p):
    rng = np.random.default_rng(p)
    img = rng.integers(0, 256, (40, 50)).astype(np.int64)
    t = rng.integers(0, 256, (p, p)).astype(np.int64) * (rng.random((p, p)) < 0.7)
    want = signal.correlate2d(img, t, mode="valid")
    assert np.array_equal(cvr.match_template_ccorr_exact(img, t), want)


def test_window_sum_with_top_left_anchor_and_zero_border():
    rng = np.random.default_rng(3)
    m = (rng.random((30, 41)) < 0.1).astype(np.int64)
    for ph, pw in ((9, 9), (5, 7), (15, 15)):
        padded = np.zeros((30 + ph - 1, 41 + pw - 1), np.int64)
        padded[:30, :41] = m
        want = signal.correlate2d(padded, np.ones((ph, pw), np.int64), mode="valid")
        assert np.array_equal(cvr.window_sum_topleft(m.astype(np.uint8), ph, pw), want)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_box_dilation_window(k):
    """A k x k box whose window runs from -k//2 to k-1-k//2 around the pixel (where OpenCV's default
    anchor, k//2, puts it) is SciPy's maximum filter of size k with origin 0 and a zero border."""
    rng = np.random.default_rng(20 + k)
    img = rng.integers(0, 60, (64, 80, 3), dtype=np.uint8)
    grey = cvr.bgr2gray_u8(img)
    m = (grey <= 10).astype(np.uint8)
    m[:10, :] = 0
    m[-10:, :] = 0
    m[:, :10] = 0
    m[:, -10:] = 0
    want = ndimage.maximum_filter(m, size=k, mode="constant", cval=0, origin=0)
    assert np.array_equal(pipeline_mask(img, dilation_size=k), want)


def test_np_gradient_matches_hand_formula():
    g = np.random.default_rng(4).integers(0, 256, (12, 17)).astype(np.uint8)
    gy, gx = np.gradient(g)
    f = g.astype(np.float64)
    assert np.array_equal(gy[1:-1], (f[2:] - f[:-2]) / 2.0) and np.array_equal(gy[0], f[1] - f[0])
    assert np.array_equal(gx[:, 1:-1], (f[:, 2:] - f[:, :-2]) / 2.0) and np.array_equal(gx[:, -1], f[:, -1] - f[:, -2])
