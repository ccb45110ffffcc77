"""CPU tests of the oracle: hand-derivable known answers for each restated primitive, NumPy
anchoring of the summation order, NumPy-oracle vs C-oracle agreement, committed fixtures.
The reference holds no tests or fixtures for this path (parity unpinned); every expected value
here is either derivable by hand or produced by NumPy itself."""
import numpy as np
import pytest

from helpers import golden_cases, load_golden
from oracle import c_oracle
from oracle import cv2_restated as cvr
from oracle.criminisi_np import CriminisiOracle, brute_force_costs


def test_front_single_pixel():
    m = np.zeros((5, 5), np.uint8)
    m[2, 2] = 1
    f = cvr.laplacian_ksize3_u8(m)
    exp = np.zeros((5, 5), np.uint8)
    for dy, dx in ((-1, -1), (-1, 1), (1, -1), (1, 1)):
        exp[2 + dy, 2 + dx] = 2
    assert np.array_equal(f, exp)


def test_front_reflect101_border():
    # masked pixel on row 0: (1, x+-1) see it twice (rows -1 and +1 both map to... row 1 <- reflect)
    m = np.zeros((5, 7), np.uint8)
    m[0, 3] = 1
    f = cvr.laplacian_ksize3_u8(m)
    assert f[1, 2] == 2 and f[1, 4] == 2
    # a pixel on row 0 looks at rows -1 -> 1 and +1 -> 1: its diagonals are counted twice
    m2 = np.zeros((5, 7), np.uint8)
    m2[1, 3] = 1
    f2 = cvr.laplacian_ksize3_u8(m2)
    assert f2[0, 2] == 4 and f2[0, 4] == 4 and f2[2, 2] == 2


def test_normal_vertical_edge():
    m = np.zeros((7, 8), np.uint8)
    m[:, 4:] = 1
    o = CriminisiOracle(np.zeros((7, 8, 3), np.uint8), m, 3)
    n = o._calc_normal_matrix()
    assert np.all(n[:, 3, 0] == -1) and np.all(n[:, 4, 0] == -1)
    assert np.all(n[:, 3, 1] == 0)
    assert np.all(n[:, 1, :] == 0) and np.all(n[:, 6, :] == 0)


def test_normal_isolated_pixel_guard():
    m = np.zeros((5, 5), np.uint8)
    m[2, 2] = 1
    o = CriminisiOracle(np.zeros((5, 5, 3), np.uint8), m, 3)
    n = o._calc_normal_matrix()
    assert np.all(n[2, 2] == 0) and np.all(np.isfinite(n))


def test_gradient_no_wraparound():
    g = np.array([[200, 3, 3, 250]], np.uint8).repeat(3, axis=0)
    gy, gx = np.gradient(g)
    assert gx[0, 0] == -197 and gx[0, 1] == -98.5 and gx[0, 3] == 247 and np.all(gy == 0)


def test_bgr2gray_known():
    px = np.array([[[255, 255, 255], [0, 0, 0], [0, 0, 255], [0, 255, 0], [255, 0, 0]]], np.uint8)
    assert cvr.bgr2gray_u8(px).tolist() == [[255, 0, 76, 150, 29]]


def test_lab_known_colours():
    # float-formula CIELAB of the sRGB primaries scaled to 8 bits (L*255/100, a+128, b+128)
    px = np.array([[[255, 255, 255], [0, 0, 0], [0, 0, 255], [0, 255, 0], [255, 0, 0],
                    [128, 128, 128]]], np.uint8)
    lab = cvr.bgr2lab_u8(px)[0].astype(int)
    exp = np.array([[255, 128, 128], [0, 128, 128], [136, 208, 195], [224, 42, 211],
                    [82, 207, 20], [137, 128, 128]])
    assert np.abs(lab - exp).max() <= 1
    assert cvr.SRGB_GAMMA_TAB[0] == 0 and cvr.SRGB_GAMMA_TAB[255] == 2040
    assert cvr.LAB_CBRT_TAB[2040] == 32768
    assert cvr.LAB_COEFFS.reshape(3, 3).sum(axis=1).tolist() == [4096, 4096, 4096]


def test_lab_greys_are_neutral():
    g = np.arange(256, dtype=np.uint8)
    px = np.repeat(g[None, :, None], 3, axis=2)
    lab = cvr.bgr2lab_u8(px)[0]
    assert np.all(lab[:, 1] == 128) and np.all(lab[:, 2] == 128)
    assert np.all(np.diff(lab[:, 0].astype(int)) >= 0)


def test_cbrt_polynomial_accuracy():
    xs = np.linspace(0.001, 1.5, 4001).astype(np.float32)
    err = [abs(float(cvr.opencv_cbrt_f32(x)) - float(x) ** (1 / 3)) / float(x) ** (1 / 3) for x in xs]
    assert max(err) < 2.0 ** -22


def _pairwise_model(a):
    n = len(a)
    if n < 8:
        r = -0.0
        for v in a:
            r = r + v
        return r
    if n <= 128:
        r = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = r[j] + a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise_model(a[:n2]) + _pairwise_model(a[n2:])


@pytest.mark.parametrize("shape", [(9, 9), (5, 9), (9, 5), (5, 5), (7, 7), (11, 11), (15, 15),
                                   (3, 3), (2, 3), (15, 8), (31, 31), (17, 20)])
def test_confidence_sum_order_matches_numpy(shape):
    """The flattened-pairwise model used by the C oracle and the GPU equals np.sum(np.sum(view))
    on non-contiguous views, bit for bit (criminisi_inpainter.py:124-125)."""
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    c = rng.random((64, 64)) ** 3
    for _ in range(100):
        y = rng.integers(0, 64 - shape[0] + 1)
        x = rng.integers(0, 64 - shape[1] + 1)
        s = c[y:y + shape[0], x:x + shape[1]]
        assert _pairwise_model(list(s.ravel())) == np.sum(np.sum(s))


def test_confidence_known_values():
    m = np.zeros((20, 20), np.uint8)
    m[10:, :] = 1
    o = CriminisiOracle(np.zeros((20, 20, 3), np.uint8), m, 9)
    o._find_front()
    o._update_confidence()
    # row 9 is the front; interior front pixel: 5 known rows of 9 -> 45/81
    assert o.confidence[9, 10] == 45.0 / 81.0
    # clipped at the left border: 5 rows x 5 columns known
    assert o.confidence[9, 0] == 25.0 / 81.0


def test_search_decomposed_equals_brute_force():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (30, 34, 3), dtype=np.uint8)
    m = np.zeros((30, 34), np.uint8)
    m[12:18, 14:22] = 1
    m[0:3, 0:4] = 1
    o = CriminisiOracle(img, m, 7)
    o._find_front()
    o._update_priorities()
    for tp in [o._find_highest_priority_pixel(), (3, 4), (2, 1)]:
        a, _, _ = o._search_costs(tp)
        b = brute_force_costs(o, tp)
        assert np.array_equal(a, b)


def test_search_constant_image_nearest_window():
    img = np.full((24, 24, 3), 77, np.uint8)
    m = np.zeros((24, 24), np.uint8)
    m[10:14, 10:14] = 1
    o = CriminisiOracle(img, m, 5)
    o.step()
    tr, tc, sy, sx, cost, rem = o.trace[0]
    costs_all_equal_ssd = cost - np.floor(cost)  # only the distance term differs between windows
    assert 0 <= costs_all_equal_ssd < 24 * 2
    # chosen window holds no masked pixel
    assert not m[sy:sy + 5, sx:sx + 5].any()


def test_search_all_rejected_returns_origin():
    img = np.random.default_rng(0).integers(0, 256, (12, 12, 3), dtype=np.uint8)
    m = np.ones((12, 12), np.uint8)
    m[5, 5] = 0
    o = CriminisiOracle(img, m, 9)
    o.step()
    assert o.trace[0][2:4] == (0, 0) and o.trace[0][4] == np.inf


def test_loop_empty_mask_and_single_pixel():
    img = np.random.default_rng(1).integers(0, 256, (16, 16, 3), dtype=np.uint8)
    o = CriminisiOracle(img, np.zeros((16, 16), np.uint8), 9)
    out = o.inpaint()
    assert np.array_equal(out, img) and len(o.trace) == 1
    m = np.zeros((16, 16), np.uint8)
    m[8, 8] = 1
    o = CriminisiOracle(img, m, 9)
    out = o.inpaint()
    assert len(o.trace) == 1 and np.array_equal(out[m == 0], img[m == 0])


def test_update_only_touches_masked_pixels():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    m = np.zeros((32, 32), np.uint8)
    m[10:20, 12:22] = 1
    o = CriminisiOracle(img, m, 9)
    o.step()
    changed = np.any(o.working_image != img, axis=2)
    assert not changed[m == 0].any()
    tr, tc = o.trace[0][:2]
    filled = (m == 1) & (o.working_mask == 0)
    assert filled.any() and np.all(o.confidence[filled] == o.confidence[tr, tc])


def _random_case(rng, k):
    H = int(rng.integers(20, 64))
    W = int(rng.integers(20, 64))
    p = int(rng.choice([3, 4, 5, 7, 9, 11, 15]))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if k % 2 else \
        np.repeat(rng.integers(0, 256, (H, W, 1), dtype=np.uint8), 3, axis=2)
    m = np.zeros((H, W), np.uint8)
    for _ in range(3):
        y0 = int(rng.integers(0, H - 4))
        x0 = int(rng.integers(0, W - 4))
        m[y0:y0 + int(rng.integers(2, H // 3)), x0:x0 + int(rng.integers(2, W // 3))] = 1
    return img, m, p


def test_c_oracle_matches_numpy_oracle():
    rng = np.random.default_rng(77)
    for k in range(10):
        img, m, p = _random_case(rng, k)
        o = CriminisiOracle(img, m, p)
        out = o.inpaint()
        cout, cconf, cmask, tr, n = c_oracle.run(img, m, p)
        assert n == len(o.trace)
        assert np.array_equal(out, cout)
        assert np.array_equal(o.confidence, cconf)
        assert not cmask.any()
        assert [(a, b, c, d) for a, b, c, d, e, f in o.trace] == \
            list(zip(tr["tr"].tolist(), tr["tc"].tolist(), tr["sy"].tolist(), tr["sx"].tolist()))
        assert np.array_equal(np.array([e for *_, e, f in o.trace]), tr["cost"])
        assert [f for *_, f in o.trace] == tr["remaining"].tolist()


@pytest.mark.parametrize("name", golden_cases())
def test_c_oracle_reproduces_golden(name):
    g = load_golden(name)
    out, conf, msk, tr, n = c_oracle.run(g["image"], g["mask"], int(g["patch_size"]))
    assert n == len(g["cost"])
    assert np.array_equal(out, g["out"])
    assert np.array_equal(conf, g["confidence"])
    assert np.array_equal(np.stack([tr["tr"], tr["tc"], tr["sy"], tr["sx"]], 1), g["trace"])
    assert np.array_equal(tr["cost"], g["cost"])
    assert np.array_equal(tr["remaining"], g["remaining"])


def test_dataset_fixture_matches_the_oracle_and_the_decoder():
    """tests/golden/xray_dataset_expected.json was made by the C oracle: one image is recomputed
    here; the decoded pixels of all 100 files must be the ones the fixtures were made from."""
    import json
    import os
    import zlib
    import crim_batch_driver as drv
    from helpers import GOLDEN, pipeline_mask
    from oracle import c_oracle
    src = os.path.join(GOLDEN, "xray_images")
    with open(os.path.join(GOLDEN, "xray_dataset_expected.json")) as f:
        want = json.load(f)
    with open(os.path.join(src, "MANIFEST.json")) as f:
        manifest = json.load(f)
    assert sorted(want) == sorted(manifest) and len(want) == 100
    for n in sorted(want):
        bgr = drv.read_bgr(os.path.join(src, n))
        assert bgr.shape == (256, 256, 3)
        assert zlib.crc32(bgr.tobytes()) == manifest[n]["crc32_bgr"], n
    n = "im037-healthy.jpg" if "im037-healthy.jpg" in want else sorted(want)[36]
    bgr = drv.read_bgr(os.path.join(src, n))
    m = pipeline_mask(bgr, dilation_size=4)
    assert int(m.sum()) == want[n]["mask_pixels"]
    out, conf, _, tr, it = c_oracle.run(bgr, m, 9)
    assert it == want[n]["iterations"]
    assert zlib.crc32(out.tobytes()) == want[n]["crc32_out"]
    assert zlib.crc32(conf.tobytes()) == want[n]["crc32_conf"]
    assert zlib.crc32(np.ascontiguousarray(tr).tobytes()) == want[n]["crc32_trace"]


def test_search_alone_and_replay_agree_with_the_full_loop():
    import crim_synth
    from oracle import c_oracle
    img = crim_synth.xray_like(48, 52, 77, colour=True)
    m = crim_synth.blob_mask(48, 52, 77, fraction=0.08, margin=7)
    out, conf, _, tr, n = c_oracle.run(img, m, 7)
    for k in (0, n // 2, n - 1):
        simg, smsk = c_oracle.replay(img, m, 7, tr, k)
        sy, sx, cost = c_oracle.search(simg, smsk, 7, tr["tr"][k], tr["tc"][k])
        assert (sy, sx, cost) == (tr["sy"][k], tr["sx"][k], tr["cost"][k])
    rimg, rmsk = c_oracle.replay(img, m, 7, tr, n)
    assert np.array_equal(rimg, out) and not rmsk.any()
