"""Pins for the restated OpenCV primitives (oracle/cv2_restated.py, tests/helpers.pipeline_mask)
against the real library.  OpenCV is absent from the build container and the GPU boxes, so these
tests SKIP there; they run wherever `import cv2` works and then pin, primitive by primitive, what
the oracle otherwise only restates from OpenCV's published algorithms.  Each test uses the call,
flags and operand types of the reference's own call site (cited)."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from helpers import pipeline_mask  # noqa: E402
from oracle import cv2_restated as cvr  # noqa: E402


def _masks():
    rng = np.random.default_rng(5)
    out = []
    for h, w in ((5, 5), (16, 23), (40, 37), (1, 9), (9, 1), (2, 2)):
        m = (rng.random((h, w)) < 0.3).astype(np.uint8)
        out.append(m)
        e = np.zeros((h, w), np.uint8)
        e[0, :] = 1
        e[:, -1] = 1
        out.append(e)
    return out


def test_laplacian_ksize3_cv8u():
    # criminisi_inpainter.py:103
    for m in _masks():
        if min(m.shape) < 2:
            continue
        assert np.array_equal(cv2.Laplacian(m, cv2.CV_8U, ksize=3), cvr.laplacian_ksize3_u8(m)), m.shape


def test_filter2d_normal_kernels_f64():
    # criminisi_inpainter.py:140-150
    xk = np.array([[.25, 0, -.25], [.5, 0, -.5], [.25, 0, -.25]])
    yk = np.array([[-.25, -.5, -.25], [0, 0, 0], [.25, .5, .25]])
    for m in _masks():
        if min(m.shape) < 2:
            continue
        for k in (xk, yk):
            assert np.array_equal(cv2.filter2D(m.astype(float), -1, k), cvr.filter2d_3x3_f64(m.astype(float), k))


def _colours():
    rng = np.random.default_rng(6)
    greys = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, axis=1)
    rnd = rng.integers(0, 256, (65536, 3), dtype=np.uint8)
    corners = np.array([[b, g, r] for b in (0, 1, 127, 128, 254, 255) for g in (0, 1, 127, 128, 254, 255)
                        for r in (0, 1, 127, 128, 254, 255)], np.uint8)
    return np.concatenate([greys, rnd, corners])[None, :, :]


def test_bgr2gray_u8():
    # criminisi_inpainter.py:170, utils.py:79
    c = _colours()
    assert np.array_equal(cv2.cvtColor(c, cv2.COLOR_BGR2GRAY), cvr.bgr2gray_u8(c))


def test_bgr2lab_u8_including_every_grey():
    # criminisi_inpainter.py:225 -- pins the rounding of the gamma and cube-root tables
    c = _colours()
    assert np.array_equal(cv2.cvtColor(c, cv2.COLOR_BGR2LAB), cvr.bgr2lab_u8(c))


def test_bgr2lab_u8_exhaustive_on_a_lattice():
    v = np.arange(0, 256, 5, dtype=np.uint8)
    b, g, r = np.meshgrid(v, v, v, indexing="ij")
    c = np.stack([b.ravel(), g.ravel(), r.ravel()], axis=1)[None]
    assert np.array_equal(cv2.cvtColor(c, cv2.COLOR_BGR2LAB), cvr.bgr2lab_u8(c))


@pytest.mark.parametrize("p", [7, 9, 15])
def test_match_template_ccorr_terms_are_the_exact_integers(p):
    # criminisi_inpainter.py:237-247: float32 planes holding integers, template = 1 - mask patch
    rng = np.random.default_rng(p)
    img = rng.integers(0, 256, (60, 70, 3), dtype=np.uint8)
    lab = cv2.cvtColor(img, cv2.COLOR_BGR2LAB).astype(np.float32)
    k = (rng.random((p, p)) < 0.7).astype(np.float32)
    total = np.zeros((60 - p + 1, 70 - p + 1), np.float64)
    exact = np.zeros_like(total)
    for ch in range(3):
        I = lab[:, :, ch]
        T = lab[20:20 + p, 30:30 + p, ch]
        tm_T = (k * T).astype(np.float32)
        term1 = cv2.matchTemplate((I ** 2).astype(np.float32), k, cv2.TM_CCORR)
        term2 = cv2.matchTemplate(I.astype(np.float32), tm_T, cv2.TM_CCORR)
        total += term1 - 2 * term2
        Ii = I.astype(np.int64)
        exact += (cvr.match_template_ccorr_exact(Ii * Ii, k.astype(np.int64))
                  - 2 * cvr.match_template_ccorr_exact(Ii, tm_T.astype(np.int64)))
    # the library may go through an FFT: the values agree to rounding noise far below the
    # smallest gap (1) between two exact sums; away from zero they are equal
    assert np.max(np.abs(total - exact)) < 0.5
    assert np.array_equal(np.rint(total), exact)


def test_window_mask_sum_anchor_topleft_border_constant():
    # criminisi_inpainter.py:250-255
    rng = np.random.default_rng(8)
    m = (rng.random((40, 52)) < 0.05).astype(np.uint8)
    for ph, pw in ((9, 9), (5, 9), (15, 15), (3, 3)):
        got = cv2.filter2D(m.astype(np.float32), -1, np.ones((ph, pw), np.float32), anchor=(0, 0),
                           borderType=cv2.BORDER_CONSTANT)
        want = cvr.window_sum_topleft(m, ph, pw)
        assert np.array_equal(np.rint(got).astype(np.int64), want)
        # the reference tests `> 0`; a DFT-based filter2D may leave +-1e-6 noise on clean windows
        clean = want == 0
        assert np.all(np.abs(got[clean]) < 1e-3)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_mask_recipe_of_inpaint_image(k):
    # utils.py:79-100: BGR2GRAY, inRange(0, 10), 10-pixel frame, dilate k x k, binarise
    rng = np.random.default_rng(9 + k)
    img = rng.integers(0, 60, (64, 80, 3), dtype=np.uint8)
    grey = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
    m = cv2.inRange(grey, 0, 10)
    m[:10, :] = 0
    m[-10:, :] = 0
    m[:, :10] = 0
    m[:, -10:] = 0
    m = cv2.dilate(m, np.ones((k, k), np.uint8), iterations=1)
    m[m > 0] = 1
    assert np.array_equal(m, pipeline_mask(img, dilation_size=k))


def test_numpy_oracle_iteration_with_real_opencv_primitives(monkeypatch):
    """The oracle's loop with every restated primitive swapped for the real cv2 call gives the
    same trace and output on a small case (exact-integer SSD against cv2's float path: the
    arg-min may only differ where two costs are closer than the FFT noise, which this case
    avoids by construction -- noise image, distinct costs)."""
    from oracle.criminisi_np import CriminisiOracle
    rng = np.random.default_rng(10)
    img = rng.integers(0, 256, (40, 44, 3), dtype=np.uint8)
    m = np.zeros((40, 44), np.uint8)
    m[15:24, 18:30] = 1
    want = CriminisiOracle(img, m, 9)
    want_out = want.inpaint()
    monkeypatch.setattr(cvr, "laplacian_ksize3_u8", lambda x: cv2.Laplacian(x, cv2.CV_8U, ksize=3))
    monkeypatch.setattr(cvr, "filter2d_3x3_f64", lambda s, k: cv2.filter2D(s, -1, np.asarray(k, dtype=np.float64)))
    monkeypatch.setattr(cvr, "bgr2gray_u8", lambda x: cv2.cvtColor(x, cv2.COLOR_BGR2GRAY))
    monkeypatch.setattr(cvr, "bgr2lab_u8", lambda x: cv2.cvtColor(x, cv2.COLOR_BGR2LAB))
    got = CriminisiOracle(img, m, 9)
    got_out = got.inpaint()
    assert got.trace == want.trace
    assert np.array_equal(got_out, want_out)


# ---- the noise-filter stage (utils.py:106-138), restated in oracle/noise_filter_np.py -----------

def _nf():
    from oracle import noise_filter_np as nf
    return nf


@pytest.mark.parametrize("k", [3, 5, 7])
def test_median_blur(k):
    rng = np.random.default_rng(k)
    img = rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)
    assert np.array_equal(cv2.medianBlur(img, k), _nf().median_blur(img, k))


def test_linear_bgr_to_lab_and_back():
    c = _colours()
    lab = cv2.cvtColor(c, cv2.COLOR_LBGR2Lab)
    assert np.array_equal(lab, _nf().lbgr2lab_u8(c))
    # every (L, a, b) byte triple on a lattice, not only those a colour maps to
    v = np.arange(0, 256, 5, dtype=np.uint8)
    L, a, b = np.meshgrid(v, v, v, indexing="ij")
    t = np.stack([L.ravel(), a.ravel(), b.ravel()], axis=1)[None]
    assert np.array_equal(cv2.cvtColor(t, cv2.COLOR_Lab2LBGR), _nf().lab2lbgr_u8(t))
    assert np.array_equal(cv2.cvtColor(lab, cv2.COLOR_Lab2LBGR), _nf().lab2lbgr_u8(lab))


@pytest.mark.parametrize("h", [3, 7, 15])
def test_fast_nl_means_one_and_two_channels(h):
    rng = np.random.default_rng(h)
    one = rng.integers(60, 200, (48, 40), dtype=np.uint8)
    assert np.array_equal(cv2.fastNlMeansDenoising(one, None, h=h, templateWindowSize=9, searchWindowSize=31),
                          _nf().fast_nl_means_u8(one, h))
    two = rng.integers(100, 160, (40, 44, 2), dtype=np.uint8)
    assert np.array_equal(cv2.fastNlMeansDenoising(two, None, h=1, templateWindowSize=9, searchWindowSize=31),
                          _nf().fast_nl_means_u8(two, 1))


def test_fast_nl_means_colored_and_the_whole_stage():
    import crim_synth
    img = crim_synth.xray_like(64, 72, 77, colour=True)
    got = cv2.fastNlMeansDenoisingColored(img, None, h=7, hColor=1, templateWindowSize=9, searchWindowSize=31)
    assert np.array_equal(got, _nf().fast_nl_means_colored(img, 7, 1))
    # filter_noise, utils.py:106-138, call for call
    x = cv2.medianBlur(img, 3)
    x = cv2.fastNlMeansDenoisingColored(x, None, h=7, hColor=1, templateWindowSize=9, searchWindowSize=31)
    lap = cv2.filter2D(x, cv2.CV_8U, np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]]))
    x = cv2.subtract(x, lap)
    assert np.array_equal(x, _nf().filter_noise(img, 3, 7, True))
