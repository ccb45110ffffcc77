"""The noise-filter stage (reference: filter_noise, src/utils.py:106-138).  CPU part: known-answer
tests of the NumPy restatement (oracle/noise_filter_np.py; PARITY UNPINNED -- OpenCV is absent, see
that file).  GPU part: the HIP kernels through the C ABI against the restatement, bit for bit."""
import numpy as np
import pytest

import crim_synth
from oracle import noise_filter_np as nf


def test_median_known_answers():
    img = np.zeros((5, 5, 3), np.uint8)
    img[2, 2] = 255                                   # an isolated bright pixel disappears
    assert not nf.median_blur(img, 3).any()
    ramp = np.repeat(np.arange(7, dtype=np.uint8)[None, :, None] * 10, 3, axis=2).repeat(4, axis=0)
    assert np.array_equal(nf.median_blur(ramp, 3), ramp)   # replicate border keeps a ramp's ends
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    for k in (3, 5, 7):
        r = k // 2
        pad = np.pad(x, ((r, r), (r, r), (0, 0)), mode="edge")
        want = np.array([[[np.median(pad[i:i + k, j:j + k, c]) for c in range(3)] for j in range(11)]
                         for i in range(9)]).astype(np.uint8)
        assert np.array_equal(nf.median_blur(x, k), want)


def test_linear_lab_round_trip_and_greys():
    greys = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    lab = nf.lbgr2lab_u8(greys)
    assert np.all(lab[:, :, 1:] == 128)                           # greys have no chroma
    assert lab[0, 0, 0] == 0 and lab[0, 255, 0] == 255 and np.all(np.diff(lab[0, :, 0].astype(int)) >= 0)
    assert lab[0, 128, 0] == 194                                  # 2.55 * (116 * (128/255)^(1/3) - 16)
    assert np.abs(nf.lab2lbgr_u8(lab).astype(int) - greys).max() <= 1
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)
    assert np.abs(nf.lab2lbgr_u8(nf.lbgr2lab_u8(img)).astype(int) - img).max() <= 4


def test_nlm_weight_table_known_values():
    tab = nf.nlm_weight_table(7, 1)
    assert nf.nlm_fixed_point_mult(31) == 8763 and nf.nlm_shift(9) == 7
    assert len(tab) == int(255 * 255 / (128 / 81) + 1) and tab[0] == 8763
    assert tab[1] == round(8763 * np.exp(-(128 / 81) / 49))
    nz = np.nonzero(tab)[0]
    assert nz[-1] == nz.size - 1                                   # the weights are a non-zero prefix ...
    assert tab[nz[-1]] >= 0.001 * 8763 > 8763 * np.exp(-((nz[-1] + 1) * 128 / 81) / 49) - 0.5   # ... cut at 0.001
    assert np.array_equal(nf.nlm_weight_table(1, 2)[:10], [8763, 3977, 1805, 819, 372, 169, 77, 35, 16, 0])


def test_nlm_constant_and_tiny_h_and_brute_force():
    c = np.full((12, 13), 91, np.uint8)
    assert np.array_equal(nf.fast_nl_means_u8(c, 7), c)
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, (14, 15), dtype=np.uint8)
    assert np.array_equal(nf.fast_nl_means_u8(x, 0.05), x)        # only the pixel itself keeps a weight
    # brute force on a tiny 2-channel image with small windows
    y = rng.integers(100, 110, (7, 8, 2), dtype=np.uint8)
    tw, sw, h = 3, 5, 4.0
    got = nf.fast_nl_means_u8(y, h, tw, sw)
    b = tw // 2 + sw // 2
    ext = nf._reflect101_pad(y, b).astype(np.int64)
    tab = nf.nlm_weight_table(h, 2, tw, sw)
    want = np.zeros_like(y)
    for i in range(7):
        for j in range(8):
            est = np.zeros(2, np.int64)
            ws = 0
            for dy in range(-(sw // 2), sw // 2 + 1):
                for dx in range(-(sw // 2), sw // 2 + 1):
                    d = 0
                    for u in range(-(tw // 2), tw // 2 + 1):
                        for v in range(-(tw // 2), tw // 2 + 1):
                            d += int(((ext[b + i + u, b + j + v] - ext[b + i + dy + u, b + j + dx + v]) ** 2).sum())
                    w = int(tab[d >> nf.nlm_shift(tw)])
                    est += w * ext[b + i + dy, b + j + dx]
                    ws += w
            want[i, j] = (est + ws // 2) // ws
    assert np.array_equal(got, want)


def test_sharpen_known_answers():
    flat = np.full((6, 7, 3), 50, np.uint8)
    assert np.array_equal(nf.laplacian_sharpen(flat), flat)
    img = flat.copy()
    img[3, 3] = 90                                    # a bright pixel: its Laplacian is negative -> 0
    out = nf.laplacian_sharpen(img)
    assert np.all(out[3, 3] == 90)
    assert np.all(out[2, 3] == 50 - 40) and np.all(out[3, 2] == 10)   # neighbours: +40 response subtracted
    img[3, 3] = 0                                     # a dark pixel: Laplacian 4 * 50 = 200 -> 0 - 200 saturates
    assert np.all(nf.laplacian_sharpen(img)[3, 3] == 0)


# ---------------------------------------------------------------------------------------------

gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("shape,k,h,sharpen,colour", [
    ((48, 56), 3, 7, True, True), ((40, 33), 5, 3, False, True), ((64, 64), 7, 15, True, False),
    ((17, 70), 3, 10, True, True), ((20, 9), 1, 5, True, True)])
def test_gpu_filter_noise_matches_restatement(shape, k, h, sharpen, colour):
    import crim_hip_capi as capi
    img = crim_synth.xray_like(shape[0], shape[1], 500 + k + h, colour=colour)
    rng = np.random.default_rng(k)
    img[rng.random(shape) < 0.02] = 255                 # salt for the median
    got = capi.filter_noise(img, median_ksize=k, nlm_h=h, apply_sharpening=sharpen)
    want = nf.filter_noise(img, median_ksize=k, nlm_h=h, apply_sharpening=sharpen)
    assert got.shape == img.shape and got.dtype == np.uint8
    assert np.array_equal(got, want)


@gpu
def test_gpu_filter_noise_stage_by_stage_on_noise():
    import crim_hip_capi as capi
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (36, 44, 3), dtype=np.uint8)
    # median only: h so small that only the pixel itself has a weight, no sharpening; what is left
    # is the median and the Lab round trip
    got = capi.filter_noise(img, median_ksize=3, nlm_h=0.01, apply_sharpening=False)
    want = nf.lab2lbgr_u8(nf.lbgr2lab_u8(nf.median_blur(img, 3)))
    assert np.array_equal(got, want)
    got = capi.filter_noise(img, median_ksize=1, nlm_h=0.01, apply_sharpening=True)
    assert np.array_equal(got, nf.laplacian_sharpen(nf.lab2lbgr_u8(nf.lbgr2lab_u8(img))))


@gpu
def test_gpu_filter_noise_batch_and_engine_in_place():
    import crim_hip_capi as capi
    from oracle import c_oracle
    imgs = np.stack([crim_synth.xray_like(40, 48, 600 + i, colour=True) for i in range(3)])
    got = capi.filter_noise(imgs, median_ksize=3, nlm_h=7)
    for i in range(3):
        assert np.array_equal(got[i], nf.filter_noise(imgs[i], 3, 7, True)), i
    # inpaint, then denoise on the device, as process_image orders the stages (utils.py:216-220)
    m = crim_synth.blob_mask(40, 48, 601, fraction=0.05, margin=9)
    with capi.Engine(40, 48, 9) as eng:
        eng.upload(imgs[1], m)
        eng.run()
        eng.filter_noise(median_ksize=3, nlm_h=7, apply_sharpening=True)
        out = eng.download()[0][0]
    assert np.array_equal(out, nf.filter_noise(c_oracle.run(imgs[1], m, 9)[0], 3, 7, True))


@gpu
@pytest.mark.parametrize("h", [7, 40])
def test_gpu_nlm_kernel_generations_agree(monkeypatch, h):
    """The three forms of the non-local means (window differences; byte correlation against box-filtered
    squares; the same with row products shared down a column, the default) on an image of several tiles
    with ragged edges: identical outputs, and equal to the restatement on a crop-sized image.  h = 40
    makes the L weight table's non-zero prefix longer than the kernels' LDS copy (the default form then
    hands over to the second one)."""
    import crim_hip_capi as capi
    img = crim_synth.xray_like(150, 301, 77, colour=True)
    rng = np.random.default_rng(3)
    img = np.clip(img.astype(np.int16) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    outs = {}
    for name, env in (("shared_rows", {}), ("correlation", {"CRIM_NLM_CORR": "1"}), ("differences", {"CRIM_NLM_DIRECT": "1"})):
        for k in ("CRIM_NLM_CORR", "CRIM_NLM_DIRECT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[name] = capi.filter_noise(img, median_ksize=3, nlm_h=h, apply_sharpening=True)
    assert np.array_equal(outs["shared_rows"], outs["correlation"])
    assert np.array_equal(outs["shared_rows"], outs["differences"])
    for k in ("CRIM_NLM_CORR", "CRIM_NLM_DIRECT"):
        monkeypatch.delenv(k, raising=False)
    small = np.ascontiguousarray(img[:40, :140])
    assert np.array_equal(capi.filter_noise(small, median_ksize=3, nlm_h=h, apply_sharpening=True),
                          nf.filter_noise(small, median_ksize=3, nlm_h=h, apply_sharpening=True))
