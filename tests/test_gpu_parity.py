"""GPU parity tests: the HIP engine, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact -- identical (target, source) per iteration, identical fp64
cost and confidence, identical output bytes.  (The oracle itself is "parity unpinned" with
respect to the reference; see oracle/cv2_restated.py.)"""
import os

import numpy as np
import pytest

from helpers import golden_cases, load_golden

import crim_hip_capi as capi
import crim_synth
from oracle import c_oracle
from oracle.criminisi_np import CriminisiOracle

pytestmark = pytest.mark.gpu


def engine_run(img, mask, p, flags=0, max_iters=0):
    H, W = mask.shape
    with capi.Engine(H, W, p, flags=flags) as eng:
        eng.upload(img, mask)
        eng.run(max_iters)
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        tr = eng.trace(0)
    return out[0], conf[0], msk[0], tr


def assert_same_as_oracle(img, mask, p, flags=0):
    out, conf, msk, tr = engine_run(img, mask, p, flags)
    oout, oconf, omask, otr, n = c_oracle.run(img, mask, p)
    assert len(tr) == n, "iteration count %d vs oracle %d" % (len(tr), n)
    for f in ("tr", "tc", "sy", "sx", "remaining"):
        bad = np.nonzero(tr[f] != otr[f])[0]
        assert bad.size == 0, "%s differs first at iteration %d: %s vs %s" % (
            f, bad[0], tr[bad[0]], otr[bad[0]])
    assert np.array_equal(tr["cost"], otr["cost"])
    assert np.array_equal(out, oout)
    assert np.array_equal(conf, oconf)
    assert not msk.any()
    return n


def rand_case(seed, H, W, noise=True, colour=True, border=False):
    rng = np.random.default_rng(seed)
    if noise:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    else:
        img = crim_synth.xray_like(H, W, seed, colour=colour)
    m = np.zeros((H, W), np.uint8)
    lo = 0 if border else 10
    for _ in range(3):
        h = int(rng.integers(3, max(4, H // 4)))
        w = int(rng.integers(3, max(4, W // 4)))
        y0 = int(rng.integers(lo, H - lo - h + 1))
        x0 = int(rng.integers(lo, W - lo - w + 1))
        m[y0:y0 + h, x0:x0 + w] = 1
    return img, m


@pytest.mark.parametrize("p", [3, 5, 7, 9, 11, 13, 15])
def test_fast_kernels_interior_mask(p):
    img, m = rand_case(100 + p, 72, 88, noise=(p % 4 == 1), colour=True)
    assert_same_as_oracle(img, m, p)


@pytest.mark.parametrize("p", [3, 7, 9, 15])
def test_border_clipped_patches(p):
    img, m = rand_case(200 + p, 64, 70, noise=False, colour=True, border=True)
    m[0:6, 0:9] = 1
    m[-5:, -7:] = 1
    m[30:36, 0:4] = 1
    assert_same_as_oracle(img, m, p)


@pytest.mark.parametrize("p", [3, 5, 7, 9, 11, 13, 15])
def test_grey_images_take_the_three_plane_search(p):
    # a = b = 128 for every grey pixel: the matrix-core search stages 3 planes instead of 5
    img, m = rand_case(150 + p, 80, 300, noise=False, colour=False)
    assert_same_as_oracle(img, m, p)


@pytest.mark.parametrize("p,colour", [(3, True), (9, True), (9, False), (15, True)])
def test_vector_dot_product_search_flag(p, colour):
    img, m = rand_case(170 + p, 72, 140, noise=False, colour=colour, border=(p == 9))
    assert_same_as_oracle(img, m, p, flags=capi.FLAG_DOT4_SEARCH)


def test_engine_switches_between_grey_and_colour_uploads():
    # the 3-plane search instance serves grey engines only: a colour upload must switch the engine
    # (and its replayed graphs) to the 5-plane instance, and a mixed batch must use it throughout
    H, W = 72, 96
    grey, mg = rand_case(190, H, W, noise=False, colour=False)
    col, mc = rand_case(191, H, W, noise=False, colour=True)
    og = c_oracle.run(grey, mg, 9)
    oc = c_oracle.run(col, mc, 9)
    with capi.Engine(H, W, 9) as eng:
        for img, m, o in ((grey, mg, og), (col, mc, oc), (grey, mg, og)):
            eng.upload(img, m)
            eng.run()
            out, conf, _ = eng.download(want_conf=True)
            assert np.array_equal(out[0], o[0]) and np.array_equal(conf[0], o[1])
            assert np.array_equal(eng.trace(0), o[3].astype(capi.ITER_REC_DTYPE))
    with capi.Engine(H, W, 9, batch=2) as eng:
        eng.upload(np.stack([grey, col]), np.stack([mg, mc]))
        eng.run()
        out, _, _ = eng.download()
        assert np.array_equal(out[0], og[0]) and np.array_equal(out[1], oc[0])


def test_wide_image_spans_several_search_tiles():
    # 600 columns: three 256-origin tiles per row of matrix-core blocks, the last one partial
    img, m = rand_case(180, 40, 600, noise=True, colour=True)
    assert_same_as_oracle(img, m, 9)
    img, m = rand_case(181, 40, 600, noise=False, colour=False, border=True)
    assert_same_as_oracle(img, m, 7)


@pytest.mark.parametrize("p,W", [(9, 247), (9, 248), (9, 255), (9, 256), (9, 263), (9, 264), (9, 271),
                                 (9, 496), (9, 503), (9, 504), (9, 512), (9, 519), (9, 520),
                                 (5, 244), (5, 259), (15, 254), (15, 270), (15, 526)])
def test_search_waves_without_inadmissible_windows(p, W):
    # A search wave whose 8 origin rows hold no inadmissible window left of the image's right edge
    # starts its accumulators from zero and removes the edge windows afterwards -- only when the edge
    # cuts at most the last of a lane's four windows, i.e. when the last 256-origin tile of a row
    # holds 240..256 admissible columns.  Widths on both sides of those limits, one small hole
    # (most waves have nothing to exclude, a few do), grey and colour.
    H = 72
    for colour in (False, True):
        img = crim_synth.xray_like(H, W, 7000 + W + p, colour=colour)
        m = np.zeros((H, W), np.uint8)
        m[30:30 + p + 2, W // 3:W // 3 + p + 3] = 1
        assert_same_as_oracle(img, m, p)


@pytest.mark.parametrize("p,tiles,H,W,border,colour", [
    (9, 8, 300, 200, False, False), (5, 3, 200, 90, True, False), (13, 8, 270, 300, False, False),
    (9, 2, 140, 600, True, False), (7, 8, 33, 64, False, False), (15, 4, 210, 120, True, False),
    (9, 8, 300, 200, False, True), (3, 5, 180, 70, True, True), (11, 3, 200, 280, False, True),
    (15, 8, 260, 100, False, True)])
def test_pipelined_search_walks_several_row_tiles(p, tiles, H, W, border, colour, monkeypatch):
    # a single small image would get one row tile per block: force the many-tiles-per-block walk
    # (chunks of `tiles` rows of tiles, the last chunk shorter, the last tile reaching below the image)
    monkeypatch.setenv("CRIM_PIPE_MIN_BLOCKS", "1")
    monkeypatch.setenv("CRIM_PIPE_T", str(tiles))
    # (three planes per tile for grey images, five for colour ones)
    img, m = rand_case(400 + p + tiles, H, W, noise=False, colour=colour, border=border)
    assert_same_as_oracle(img, m, p)


@pytest.mark.parametrize("p,colour,border", [(9, True, False), (15, True, True), (5, False, True),
                                             (13, True, False)])
def test_plain_matrix_core_search_kernel(p, colour, border, monkeypatch):
    # CRIM_PIPE_T=0 selects the non-pipelined kernel (register-staged tiles, one tile per block)
    monkeypatch.setenv("CRIM_PIPE_T", "0")
    img, m = rand_case(480 + p, 90, 300, noise=False, colour=colour, border=border)
    assert_same_as_oracle(img, m, p)


def test_pipelined_and_plain_grey_search_agree_on_a_batch(monkeypatch):
    H, W, B = 96, 80, 16
    imgs, masks = [], []
    for i in range(B):
        img, m = rand_case(430 + i, H, W, noise=(i % 3 == 0), colour=False, border=(i % 4 == 1))
        imgs.append(img if i % 3 else np.repeat(img[:, :, :1], 3, axis=2))   # noise images made grey
        masks.append(m)
    imgs, masks = np.stack(imgs), np.stack(masks)
    res = []
    for t in ("0", "8"):
        monkeypatch.setenv("CRIM_PIPE_T", t)
        monkeypatch.setenv("CRIM_PIPE_MIN_BLOCKS", "1")
        with capi.Engine(H, W, 9, batch=B) as eng:
            eng.upload(imgs, masks)
            eng.run()
            out, conf, _ = eng.download(want_conf=True)
            res.append((out.copy(), conf.copy(), [eng.trace(i).copy() for i in range(B)]))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)
    oout, oconf = c_oracle.run(imgs[5], masks[5], 9)[:2]
    assert np.array_equal(res[1][0][5], oout) and np.array_equal(res[1][1][5], oconf)


@pytest.mark.parametrize("p,colour", [(9, False), (9, True), (5, False), (7, True), (13, False)])
def test_border_clipped_targets_without_selection_in_the_search(p, colour, monkeypatch):
    """Holes that touch the image border give targets whose patch is clipped: the search then checks every
    window against the mask itself (the validity map only knows full P x P windows).  With the selection in
    its own launch (the arrangement of large batches) and hole corners in the image's last rows and columns,
    so that the last tiles hold windows beyond the candidate grid -- those must not be looked at.  Single
    images and a batch whose last image has the clipped targets."""
    monkeypatch.setenv("CRIM_SELECT_IN_SEARCH", "0")
    H, W = 70, 300
    img = crim_synth.xray_like(H, W, 8100 + p, colour=colour)
    m = np.zeros((H, W), np.uint8)
    m[H - p - 3:, W - 2 * p:] = 1          # bottom-right corner
    m[:p + 1, :p + 2] = 1                  # top-left corner
    m[30:30 + p, 120:120 + p + 1] = 1      # and an interior hole
    assert_same_as_oracle(img, m, p)
    B = 4
    imgs = np.stack([crim_synth.xray_like(H, W, 8200 + p + i, colour=colour) for i in range(B)])
    masks = np.zeros((B, H, W), np.uint8)
    for i in range(B - 1):
        masks[i, 20:20 + p + i, 40 + 30 * i:40 + 30 * i + p + 2] = 1
    masks[B - 1] = m
    with capi.Engine(H, W, p, batch=B) as eng:
        eng.upload(imgs, masks)
        eng.run()
        out, conf, _ = eng.download(want_conf=True)
        traces = [eng.trace(i).copy() for i in range(B)]
    for i in (0, B - 1):
        oout, oconf, _, otr, n = c_oracle.run(imgs[i], masks[i], p)
        assert np.array_equal(out[i], oout) and np.array_equal(conf[i], oconf), i
        assert np.array_equal(traces[i], otr.astype(capi.ITER_REC_DTYPE)), i


def test_engines_created_one_after_another_with_device_uploads():
    """An engine created where another one of the same size has just been destroyed gets that one's memory,
    stale contents included, and its creation clears state with memsets that may still be running when the call
    returns: an upload from device buffers issued right away must find them done (it once overtook them and
    lost the loop state it had just written -- on batches of thousands of 256 x 256 images, where the memsets
    take a millisecond).  Three engines in a row on such a batch, a few iterations each: every one must
    advance exactly as the first, and the first as the oracle."""
    from helpers import HipBuffers
    B, H, W, iters = 8192, 256, 256, 3
    base = [(crim_synth.xray_like(H, W, 8300 + i), crim_synth.blob_mask(H, W, 8300 + i, 0.10, 9)) for i in range(8)]
    imgs = np.stack([base[i % 8][0] for i in range(B)])
    masks = np.stack([base[i % 8][1] for i in range(B)])
    dev = HipBuffers()
    try:
        d_img, d_mask, d_out = dev.to_device(imgs), dev.to_device(masks), dev.zeros(imgs.nbytes)
        del imgs, masks
        runs = []
        for k in range(3):
            with capi.Engine(H, W, 9, batch=B, flags=capi.FLAG_NO_TRACE) as eng:
                eng.upload_device(d_img, d_mask)
                it, rem = eng.run(max_iters=iters)
                eng.download_device(d_out)
                first = dev.to_host(d_out, (8, H, W, 3))
                last = eng.download(B - 8, 8)[0]
                runs.append((it, rem, first, last))
    finally:
        dev.free_all()
    it0, rem0, first0, last0 = runs[0]
    assert it0 == iters and rem0 > 0
    for i in (0, 5):
        want = c_oracle.run(base[i][0], base[i][1], 9, max_iters=iters)[0]
        assert np.array_equal(first0[i], want) and np.array_equal(last0[i], want), i
    for k in (1, 2):
        it, rem, first, last = runs[k]
        assert (it, rem) == (it0, rem0), (k, it, rem, it0, rem0)
        assert np.array_equal(first, first0) and np.array_equal(last, last0), k


@pytest.mark.parametrize("flag", [capi.FLAG_PROFILE, capi.FLAG_PROFILE_SOLO])
def test_profile_flags_count_every_search_launch(flag):
    H, W, B = 64, 72, 4
    imgs, masks = zip(*[rand_case(460 + i, H, W, noise=False, colour=(i % 2 == 0)) for i in range(B)])
    imgs, masks = np.stack(imgs), np.stack(masks)
    with capi.Engine(H, W, 9, batch=B, flags=flag) as eng:
        eng.upload(imgs, masks)
        eng.run()
        out, _, _ = eng.download()
        ms, launches, evals = eng.profile()
        iters = [len(eng.trace(i)) for i in range(B)]
    for i in range(B):
        assert np.array_equal(out[i], c_oracle.run(imgs[i], masks[i], 9)[0])
    # an image is searched once per iteration it takes part in
    assert evals == float(sum(iters)) * (H - 8) * (W - 8)
    assert launches >= max(iters) and ms > 0.0


# patch_size 1 and 2 give a 1x1 patch around a known pixel: the reference never terminates
@pytest.mark.parametrize("p", [4, 6, 8, 17, 21, 31])
def test_generic_patch_sizes(p):
    img, m = rand_case(300 + p, 48, 52, noise=False, colour=(p % 2 == 0), border=(p > 8))
    assert_same_as_oracle(img, m, p)


@pytest.mark.parametrize("p", [5, 9])
def test_generic_kernel_flag_matches(p):
    img, m = rand_case(400 + p, 60, 64, noise=False)
    assert_same_as_oracle(img, m, p, flags=capi.FLAG_GENERIC_SEARCH)


def test_eager_mode_matches():
    img, m = rand_case(450, 60, 64, noise=False)
    assert_same_as_oracle(img, m, 9, flags=capi.FLAG_EAGER)


def test_grey_replicated_image_with_many_ties():
    # flat regions: data term = 0.001 exactly on many front pixels, cost ties between windows
    img = np.full((64, 64, 3), 90, np.uint8)
    img[:, 32:] = 160
    m = np.zeros((64, 64), np.uint8)
    m[20:44, 20:44] = 1
    assert_same_as_oracle(img, m, 9)


def test_empty_mask_runs_one_noop_iteration():
    img = crim_synth.xray_like(40, 40, 5)
    m = np.zeros((40, 40), np.uint8)
    out, conf, msk, tr = engine_run(img, m, 9)
    assert len(tr) == 1 and np.array_equal(out, img) and np.all(conf == 1.0)
    assert_same_as_oracle(img, m, 9)


def test_single_pixel_hole():
    img = crim_synth.xray_like(40, 40, 6, colour=True)
    m = np.zeros((40, 40), np.uint8)
    m[17, 23] = 1
    assert assert_same_as_oracle(img, m, 9) == 1


def test_no_admissible_window_falls_back_to_origin():
    # every 9x9 window holds a masked pixel: costs are all +inf, argmin is window (0,0)
    img = np.random.default_rng(3).integers(0, 256, (20, 20, 3), dtype=np.uint8)
    m = np.zeros((20, 20), np.uint8)
    m[4::5, 4::5] = 1
    assert_same_as_oracle(img, m, 9)


def test_mask_touching_every_border():
    img = crim_synth.xray_like(48, 48, 8, colour=True)
    m = np.zeros((48, 48), np.uint8)
    m[0, :] = 1
    m[-1, :] = 1
    m[:, 0] = 1
    m[:, -1] = 1
    m[20:28, 20:28] = 1
    assert_same_as_oracle(img, m, 9)


def test_non_terminating_input_reports_no_progress():
    # patch_size 1 and a mask on every odd-parity pixel: no known pixel has a masked diagonal
    # neighbour, so the front is empty, the target falls back to (0,0), whose 1x1 patch is
    # already known -- the reference loop (:51-83) would spin for ever; the engine reports it.
    img = np.zeros((16, 16, 3), np.uint8)
    yy, xx = np.mgrid[0:16, 0:16]
    m = ((yy + xx) % 2 == 1).astype(np.uint8)
    with capi.Engine(16, 16, 1) as eng:
        eng.upload(img, m)
        with pytest.raises(capi.CriminisiHipError) as ei:
            eng.run()
        assert ei.value.code == capi.CRIM_ERR_NO_PROGRESS
        assert eng.iterations(0) == 1


def test_stepping_equals_full_run():
    img, m = rand_case(500, 56, 60, noise=False)
    out, conf, msk, tr = engine_run(img, m, 9)
    with capi.Engine(56, 60, 9) as eng:
        eng.upload(img, m)
        steps = 0
        while True:
            it, rem = eng.run(max_iters=1)
            steps += 1
            assert it == steps
            if rem == 0:
                break
        out2, conf2, _ = eng.download(want_conf=True)
        tr2 = eng.trace(0)
    assert steps == len(tr)
    assert np.array_equal(out, out2[0]) and np.array_equal(conf, conf2[0])
    assert np.array_equal(tr, tr2)


def test_fields_match_numpy_oracle_mid_run():
    img, m = rand_case(600, 52, 58, noise=False, colour=True)
    o = CriminisiOracle(img, m, 9)
    with capi.Engine(52, 58, 9) as eng:
        eng.upload(img, m)
        for k in range(4):
            front, data, prio = eng.peek_fields(0)
            # oracle: compute the same three fields for its current state without advancing
            o._find_front()
            saved = o.confidence.copy()
            o._update_priorities()
            assert np.array_equal(front, o.front)
            fm = o.front > 0
            assert np.array_equal(data[fm], o.data[fm])
            assert np.array_equal(prio, o.priority)
            o.confidence = saved
            o.step()
            eng.run(max_iters=1)
            out, conf, msk = eng.download(want_conf=True, want_mask=True)
            assert np.array_equal(out[0], o.working_image)
            assert np.array_equal(conf[0], o.confidence)
            assert np.array_equal(msk[0], o.working_mask)


def _debug_dterm(eng, H, W, index=0):
    import ctypes
    buf = np.zeros((H, W), np.float64)
    lib = capi.load_library()
    lib.crim_debug_dterm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rc = lib.crim_debug_dterm(eng._h, int(index), buf.ctypes.data)
    assert rc == 0
    return buf


@pytest.mark.parametrize("patch,shape", [(9, (96, 112)), (5, (70, 64)), (15, (100, 90))])
def test_cached_data_terms_equal_fresh_ones(patch, shape):
    """The priority stage recomputes a front pixel's data term only in 16x16 tiles within reach of
    the patch the previous iteration filled and reads every other front pixel's from a cache.
    After each of a number of iterations: every front pixel of a tile the next priority launch will
    NOT recompute must hold, in the cache, exactly the data term a full-frame evaluation gives."""
    H, W = shape
    img = crim_synth.xray_like(H, W, 8200 + patch, colour=True)
    m = crim_synth.blob_mask(H, W, 8200 + patch, 0.18, patch)
    half = (patch - 1) // 2
    checked = 0
    with capi.Engine(H, W, patch) as eng:
        eng.upload(img, m)
        for it in range(1, 40):
            done, rem = eng.run(max_iters=1)
            if rem == 0:
                break
            tr = eng.trace(0)[-1]
            r0, c0 = max(0, int(tr["tr"]) - half), max(0, int(tr["tc"]) - half)
            r1, c1 = min(int(tr["tr"]) + half, H - 1), min(int(tr["tc"]) + half, W - 1)
            front, data, prio = eng.peek_fields(0)
            cache = _debug_dterm(eng, H, W)
            # tiles the next launch recomputes: those meeting the filled box grown by half + 1
            y0, y1, x0, x1 = r0 - 1 - half, r1 + 1 + half, c0 - 1 - half, c1 + 1 + half
            ys, xs = np.nonzero(front)
            for y, x in zip(ys, xs):
                ty, tx = (y // 16) * 16, (x // 16) * 16
                fresh = ty <= y1 and ty + 15 >= y0 and tx <= x1 and tx + 15 >= x0
                if not fresh:
                    assert cache[y, x] == data[y, x], (it, y, x)
                    checked += 1
    assert checked > 200


def _debug_state(eng, index=0):
    import ctypes
    buf = (ctypes.c_int * 16)()
    rc = capi.load_library().crim_debug_state(eng._h, int(index), buf)
    assert rc == 0
    return list(buf)


@pytest.mark.parametrize("p,off", [(9, 0), (9, 1), (9, -1), (9, 4), (9, 5), (9, -5), (9, 6), (15, 0), (15, 7), (15, 8),
                                   (15, -8), (3, 0), (3, 2), (5, -3), (13, 9)])
def test_target_patches_next_to_far_tiles(p, off):
    """Single images run two launches per iteration: the search, and one launch that updates the image AND
    evaluates the next iteration's priorities -- tiles within the patch half-width + 1 of the filled patch
    ("near") from the update applied on the fly, all others ("far") from memory, concurrently with the
    update.  Holes whose edges sit on, and a few pixels either side of, the 16-pixel tile grid put target
    patches right at the boundary between the two kinds, in every direction; a second hole keeps far tiles
    with a live front around for the whole run."""
    H, W = 112, 128
    img = crim_synth.xray_like(H, W, 7300 + 10 * p + off, colour=True)
    m = np.zeros((H, W), np.uint8)
    m[32 + off:64 + off, 48 + off:96 + off] = 1      # edges on the tile grid shifted by `off`
    m[80:96, 16:32] = 1                              # exactly one tile
    m[18:26, 100:118] = 1
    assert_same_as_oracle(img, m, p)


@pytest.mark.parametrize("p,colour,border", [(9, True, False), (15, True, True), (5, False, True), (9, False, False),
                                             (3, True, True), (7, True, False), (11, False, False), (13, True, True)])
def test_three_launch_iteration_still_matches(p, colour, border, monkeypatch):
    """CRIM_FUSED=0: the iteration as priority, search (which also selects), update -- the graph that
    batches beyond the two-launch regime replay -- on single images."""
    monkeypatch.setenv("CRIM_FUSED", "0")
    img, m = rand_case(640 + p, 80, 104, noise=False, colour=colour, border=border)
    assert_same_as_oracle(img, m, p)


def test_two_launch_iteration_on_a_batch_with_stepping():
    """A small batch (two-launch iterations), images finishing at different times, advanced a few
    iterations per call: every call starts with a full priority launch that must leave the state as the
    previous call's last launch already computed it."""
    H, W, B = 80, 96, 6
    imgs = np.stack([crim_synth.xray_like(H, W, 7400 + i, colour=(i % 3 != 0)) for i in range(B)])
    masks = np.stack([crim_synth.blob_mask(H, W, 7400 + i, fraction=0.03 + 0.02 * i, margin=9) for i in range(B)])
    masks[2] = 0
    with capi.Engine(H, W, 9, batch=B) as eng:
        eng.upload(imgs, masks)
        rem, calls = 1, 0
        while rem:
            _, rem = eng.run(max_iters=1 + calls % 5)
            calls += 1
        out, conf, _ = eng.download(want_conf=True)
        traces = [eng.trace(i) for i in range(B)]
    for i in range(B):
        oout, oconf, _, otr, n = c_oracle.run(imgs[i], masks[i], 9)
        assert np.array_equal(traces[i], otr.astype(capi.ITER_REC_DTYPE)), i
        assert np.array_equal(out[i], oout) and np.array_equal(conf[i], oconf), i


@pytest.mark.parametrize("stepwise", [True, False])
def test_tile_list_outgrowing_its_capacity_is_rebuilt(stepwise):
    """A hole that covers whole 16x16 tiles: their front counts leave zero only when the fill
    reaches them, which appends them to the tile list.  The list is padded to its capacity with
    stale duplicates first, so the first such append finds it full and k_update must rebuild it
    from the counts (one iteration per host call: the periodic rebuild of the long graphs never
    runs) -- every later target must still be the oracle's."""
    H, W, p = 80, 96, 9
    img = crim_synth.xray_like(H, W, 7100, colour=True)
    m = np.zeros((H, W), np.uint8)
    m[12:70, 14:84] = 1
    oout, oconf, _, otr, n = c_oracle.run(img, m, p)
    with capi.Engine(H, W, p) as eng:
        eng.upload(img, m)
        assert capi.load_library().crim_debug_fill_tile_list(eng._h, 0) == 0
        if stepwise:
            rem = 1
            while rem:
                _, rem = eng.run(max_iters=1)
        else:
            eng.run()
        st = _debug_state(eng)
        out, conf, _ = eng.download(want_conf=True)
        tr = eng.trace(0)
    assert np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE))
    assert np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf)
    ntiles = ((H + 15) // 16) * ((W + 15) // 16)
    assert st[4] <= 2 * ntiles              # tcount stays within the list
    assert st[10] > 0, "the in-kernel rebuild never ran"


def test_batch_of_different_masks():
    H, W, B = 64, 64, 5
    imgs = np.stack([crim_synth.xray_like(H, W, 700 + i, colour=(i % 2 == 0)) for i in range(B)])
    masks = np.stack([crim_synth.blob_mask(H, W, 700 + i, fraction=0.04 + 0.03 * i, margin=9)
                      for i in range(B)])
    masks[3] = 0            # one image with nothing to do
    with capi.Engine(H, W, 9, batch=B) as eng:
        eng.upload(imgs, masks)
        eng.run()
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        traces = [eng.trace(i) for i in range(B)]
    for i in range(B):
        oout, oconf, omask, otr, n = c_oracle.run(imgs[i], masks[i], 9)
        assert len(traces[i]) == n
        assert np.array_equal(traces[i], otr.astype(capi.ITER_REC_DTYPE))
        assert np.array_equal(out[i], oout) and np.array_equal(conf[i], oconf)


def test_engine_reuse_after_reupload():
    H, W = 48, 48
    with capi.Engine(H, W, 9) as eng:
        for seed in (801, 802):
            img = crim_synth.xray_like(H, W, seed, colour=True)
            m = crim_synth.blob_mask(H, W, seed, fraction=0.08, margin=9)
            eng.upload(img, m)
            eng.run()
            out, conf, _ = eng.download(want_conf=True)
            oout, oconf, _, otr, n = c_oracle.run(img, m, 9)
            assert np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf)
            assert np.array_equal(eng.trace(0), otr.astype(capi.ITER_REC_DTYPE))


@pytest.mark.parametrize("name", golden_cases())
def test_golden_fixture(name):
    g = load_golden(name)
    out, conf, msk, tr = engine_run(g["image"], g["mask"], int(g["patch_size"]))
    assert len(tr) == len(g["cost"])
    assert np.array_equal(np.stack([tr["tr"], tr["tc"], tr["sy"], tr["sx"]], 1), g["trace"])
    assert np.array_equal(tr["cost"], g["cost"])
    assert np.array_equal(tr["remaining"], g["remaining"])
    assert np.array_equal(out, g["out"])
    assert np.array_equal(conf, g["confidence"])


def test_shim_class_end_to_end():
    from src.criminisi_inpainter import Criminisi_Inpainter
    img = crim_synth.xray_like(64, 64, 901, colour=True)
    m = crim_synth.blob_mask(64, 64, 901, fraction=0.06, margin=9)
    img0, m0 = img.copy(), m.copy()
    ip = Criminisi_Inpainter(img, m, patch_size=9)
    out = ip.inpaint()
    oout, oconf, _, otr, n = c_oracle.run(img0, m0, 9)
    assert out.dtype == np.uint8 and out.shape == (64, 64, 3)
    assert np.array_equal(out, oout) and np.array_equal(ip.confidence, oconf)
    assert np.array_equal(img, img0) and np.array_equal(m, m0)      # inputs never mutated
    assert not ip.working_mask.any() and ip.iterations == n
    # front / data / priority: what the reference's last iteration left there (computed before
    # that iteration's update, :53-59)
    o = CriminisiOracle(img0, m0, 9)
    for _ in range(n - 1):
        o.step()
    o._find_front()
    o._update_priorities()
    assert np.array_equal(ip.front, o.front) and ip.front.any()
    assert np.array_equal(ip.priority, o.priority)
    fm = o.front > 0
    assert np.array_equal(ip.data[fm], o.data[fm])


def test_shim_verbose_and_show_progress_side_effects(tmp_path, monkeypatch, capsys):
    # verbose: the reference's lines per iteration (:67-69, :76-77, :85-87), printed in chunks;
    # show_progress: results/working_image_<c>.png before every iteration's update (:55-57)
    from src.criminisi_inpainter import Criminisi_Inpainter
    img = crim_synth.xray_like(48, 56, 902, colour=True)
    m = crim_synth.blob_mask(48, 56, 902, fraction=0.05, margin=9)
    oout, _, _, otr, n = c_oracle.run(img, m, 9)
    total = int(m.sum())
    monkeypatch.setattr(Criminisi_Inpainter, "VERBOSE_CHUNK", 3)
    import time
    t0 = time.time()
    ip = Criminisi_Inpainter(img, m, patch_size=9, verbose=True)
    out = ip.inpaint()
    wall = time.time() - t0
    assert np.array_equal(out, oout)
    lines = capsys.readouterr().out.splitlines()
    assert len(lines) == 2 * n + 1
    assert all(l.startswith('Time to find best: ') and l.endswith(' seconds') for l in lines[0:-1:2])
    # 'Time to find best' is each iteration's own search time as the device measured it (:64-69): one value
    # per iteration, positive, not all alike, together no more than the run took
    ts = ip.search_seconds
    assert ts is not None and len(ts) == n
    assert (ts > 0).all() and ts.sum() <= wall and len(set(np.round(ts * 1e8).astype(np.int64))) > 1
    assert lines[1:-1:2] == ['%d of %d completed' % (total - int(r), total) for r in otr['remaining']]
    assert lines[-1].startswith('Inpainting took ')
    monkeypatch.chdir(tmp_path)
    os.makedirs('results')
    out = Criminisi_Inpainter(img, m, patch_size=9, show_progress=True).inpaint()
    assert np.array_equal(out, oout)
    assert sorted(os.listdir('results')) == sorted('working_image_%d.png' % c for c in range(n))
    # frame c is the working image before iteration c's update
    from PIL import Image
    for c in (0, 1, n // 2, n - 1):
        frame = np.asarray(Image.open('results/working_image_%d.png' % c).convert("RGB"))[:, :, ::-1]
        want, _ = c_oracle.replay(img, m, 9, otr, c)
        assert np.array_equal(frame, want), c
    # both side effects together, chunks of 3 iterations
    for f in os.listdir('results'):
        os.remove(os.path.join('results', f))
    capsys.readouterr()
    out = Criminisi_Inpainter(img, m, patch_size=9, verbose=True, show_progress=True).inpaint()
    assert np.array_equal(out, oout)
    assert len(capsys.readouterr().out.splitlines()) == 2 * n + 1
    assert len(os.listdir('results')) == n


def test_full_size_properties_256():
    """BASELINE configs[1] shape: 256x256 grey image, 10 % mask, patch 9 -- oracle parity plus
    size-independent properties (known pixels untouched, every filled pixel is a copy of a
    pixel value present in the source region, the loop terminates with an empty mask)."""
    img = crim_synth.xray_like(256, 256, 1002)
    m = crim_synth.blob_mask(256, 256, 1002, fraction=0.10, margin=9)
    out, conf, msk, tr = engine_run(img, m, 9)
    assert not msk.any() and tr["remaining"][-1] == 0
    assert np.array_equal(out[m == 0], img[m == 0])
    assert np.all(np.diff(tr["remaining"]) < 0)
    oout, oconf, _, otr, n = c_oracle.run(img, m, 9)
    assert n == len(tr) and np.array_equal(out, oout) and np.array_equal(conf, oconf)
    assert np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE))


def test_random_stress_sweep():
    """Many small random cases (sizes, patch sizes, mask shapes incl. borders, noise / smooth /
    flat images): every one must match the oracle bit for bit."""
    rng = np.random.default_rng(31337)
    for k in range(48):
        H = int(rng.integers(20, 90))
        W = int(rng.integers(20, 90))
        p = int(rng.choice([3, 4, 5, 7, 9, 9, 9, 11, 13, 15, 17]))
        kind = k % 4
        if kind == 0:
            img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == 1:
            img = crim_synth.xray_like(H, W, 9000 + k, colour=True)
        elif kind == 2:
            img = crim_synth.xray_like(H, W, 9000 + k, colour=False)
        else:
            img = np.full((H, W, 3), int(rng.integers(0, 256)), np.uint8)
            img[:, W // 2:] = int(rng.integers(0, 256))
        m = np.zeros((H, W), np.uint8)
        for _ in range(int(rng.integers(1, 5))):
            h = int(rng.integers(1, max(2, H // 3)))
            w = int(rng.integers(1, max(2, W // 3)))
            y0 = int(rng.integers(0, H - h + 1))
            x0 = int(rng.integers(0, W - w + 1))
            m[y0:y0 + h, x0:x0 + w] = 1
        if k % 5 == 0:
            m[rng.random((H, W)) < 0.02] = 1        # salt: isolated masked pixels
        if m.all():
            m[0, 0] = 0
        try:
            assert_same_as_oracle(img, m, p)
        except AssertionError as e:
            raise AssertionError("case %d (H=%d W=%d p=%d kind=%d): %s" % (k, H, W, p, kind, e))


def test_512_prefix_and_properties():
    """BASELINE configs[2] shape (512x512x3, 10 % mask, patch 9): the first 24 iterations against
    the oracle, then the full run through size-independent properties."""
    img = crim_synth.xray_like(512, 512, 3001, colour=True)
    m = crim_synth.blob_mask(512, 512, 3001, fraction=0.10, margin=9)
    oout, oconf, omask, otr, n = c_oracle.run(img, m, 9, max_iters=24)
    with capi.Engine(512, 512, 9) as eng:
        eng.upload(img, m)
        eng.run(max_iters=24)
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        assert np.array_equal(eng.trace(0), otr.astype(capi.ITER_REC_DTYPE))
        assert np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf)
        assert np.array_equal(msk[0], omask)
        it, rem = eng.run()
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        tr = eng.trace(0)
    assert rem == 0 and not msk.any() and len(tr) == it
    assert np.array_equal(out[0][m == 0], img[m == 0])
    assert np.all(np.diff(tr["remaining"]) < 0)
    assert np.all(conf[0][m == 1] > 0) and np.all(conf[0][m == 1] <= 1)
    # every source window was free of masked pixels and inside the image
    assert tr["sy"].min() >= 0 and tr["sx"].min() >= 0
    assert tr["sy"].max() <= 512 - 9 and tr["sx"].max() <= 512 - 9


def test_512_batch_prefix_per_image():
    """BASELINE configs[2] as a BATCH: 16 false-colour 512x512 images with different 10 % masks in lock
    step (two lanes), the first iterations of every image against the oracle record for record, then the
    whole batch to completion through properties that need no oracle."""
    B, S, iters = 16, 512, 5
    imgs = np.stack([crim_synth.xray_like(S, S, 3000 + i, colour=True) for i in range(B)])
    masks = np.stack([crim_synth.blob_mask(S, S, 3000 + i, fraction=0.10, margin=9) for i in range(B)])
    with capi.Engine(S, S, 9, batch=B) as eng:
        eng.upload(imgs, masks)
        eng.run(max_iters=iters)
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        traces = [eng.trace(i) for i in range(B)]
        for i in range(B):
            oout, oconf, omask, otr, n = c_oracle.run(imgs[i], masks[i], 9, max_iters=iters)
            assert np.array_equal(traces[i][:n], otr.astype(capi.ITER_REC_DTYPE)), i
            assert np.array_equal(out[i], oout) and np.array_equal(conf[i], oconf) and np.array_equal(msk[i], omask), i
        it, rem = eng.run()
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        counts = [eng.iterations(i) for i in range(B)]
        traces = [eng.trace(i) for i in range(B)]
    assert rem == 0 and not msk.any() and len(set(counts)) > 1          # images finish at different times
    for i in range(B):
        assert np.array_equal(out[i][masks[i] == 0], imgs[i][masks[i] == 0])
        assert (np.diff(traces[i]["remaining"]) < 0).all() and traces[i]["remaining"][-1] == 0
        assert (conf[i] > 0).all() and (conf[i] <= 1).all()
        want, _ = c_oracle.replay(imgs[i], masks[i], 9, traces[i], len(traces[i]))
        assert np.array_equal(out[i], want), i


@pytest.mark.parametrize("p,iters", [(9, 4), (15, 2)])
def test_2048_prefix(p, iters):
    """BASELINE configs[3] shape (2048x2048x3, 1024x1024 hole): the first iterations against the
    oracle (a full CPU run would take hours), eight 256-origin tiles per row of search blocks."""
    S = 2048
    img = crim_synth.xray_like(S, S, 4000 + S, colour=True)
    mask = crim_synth.square_hole_mask(S, S, S // 2)
    oout, oconf, omask, otr, n = c_oracle.run(img, mask, p, max_iters=iters)
    with capi.Engine(S, S, p) as eng:
        eng.upload(img, mask)
        eng.run(max_iters=iters)
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        tr = eng.trace(0)
    assert np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE))
    assert np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf) and np.array_equal(msk[0], omask)


def check_full_run_properties(img, mask, p, out, conf, msk, tr, rem, it):
    """Size-independent properties of a completed fill (criminisi_inpainter.py:51-83, :267-308):
    the mask is empty, known pixels are untouched, every iteration fills something, every source
    window lies inside the image, confidences of filled pixels are in (0, 1], and the output is
    exactly the input with the trace's patch copies applied in order."""
    H, W = mask.shape
    assert rem == 0 and not msk.any() and len(tr) == it
    assert np.array_equal(out[mask == 0], img[mask == 0])
    assert np.all(np.diff(tr["remaining"]) < 0) and tr["remaining"][-1] == 0
    assert np.all(conf[mask == 1] > 0) and np.all(conf[mask == 1] <= 1) and np.all(conf[mask == 0] <= 1)
    assert tr["sy"].min() >= 0 and tr["sx"].min() >= 0
    assert tr["sy"].max() <= H - p and tr["sx"].max() <= W - p
    assert np.all(np.isfinite(tr["cost"]))
    rimg, rmsk = c_oracle.replay(img, mask, p, tr, len(tr))
    assert not rmsk.any() and np.array_equal(rimg, out)


def check_sampled_searches(img, mask, p, tr, picks):
    """The search of a few iterations deep inside a long run against the oracle's search on the
    state rebuilt from the trace (image and mask after the first k iterations)."""
    for k in picks:
        simg, smsk = c_oracle.replay(img, mask, p, tr, k)
        sy, sx, cost = c_oracle.search(simg, smsk, p, int(tr["tr"][k]), int(tr["tc"][k]))
        assert (sy, sx) == (int(tr["sy"][k]), int(tr["sx"][k])), "iteration %d" % k
        assert cost == tr["cost"][k], "iteration %d" % k


@pytest.mark.parametrize("p", [9, 15])
def test_2048_full_run_properties_and_sampled_searches(p):
    """BASELINE configs[3] to completion (2048x2048x3, 1024x1024 hole, patch 9 and 15): the
    size-independent properties, plus the oracle's search on three states deep inside the run."""
    S = 2048
    img = crim_synth.xray_like(S, S, 4000 + S, colour=True)
    mask = crim_synth.square_hole_mask(S, S, S // 2)
    with capi.Engine(S, S, p) as eng:
        eng.upload(img, mask)
        it, rem = eng.run()
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        tr = eng.trace(0)
    check_full_run_properties(img, mask, p, out[0], conf[0], msk[0], tr, rem, it)
    n = len(tr)
    check_sampled_searches(img, mask, p, tr, [n // 3, (2 * n) // 3, n - 1] if p == 9 else [n // 2])


def test_4096_prefix_split_and_sampled_search():
    """BASELINE configs[4] shape (4096x4096x3, 2048x2048 hole, patch 9): the first two iterations
    against the oracle, with the search as one launch and divided into 8 parts (128 rows of search
    tiles: 16 per part, several row tiles per block inside a part); then 1500 iterations both ways
    must agree record for record, and the oracle's search confirms the last of them."""
    S, p = 4096, 9
    img = crim_synth.xray_like(S, S, 4000 + S, colour=True)
    mask = crim_synth.square_hole_mask(S, S, S // 2)
    oout, oconf, omask, otr, n = c_oracle.run(img, mask, p, max_iters=2)
    traces = []
    for parts in (0, 8):
        with capi.Engine(S, S, p) as eng:
            eng.split_virtual(parts)
            eng.upload(img, mask)
            eng.run(max_iters=2)
            out, conf, msk = eng.download(want_conf=True, want_mask=True)
            assert np.array_equal(eng.trace(0), otr.astype(capi.ITER_REC_DTYPE)), parts
            assert np.array_equal(out[0], oout) and np.array_equal(msk[0], omask), parts
            assert np.array_equal(conf[0], oconf), parts
            del out, conf, msk
            eng.run(max_iters=1498)
            traces.append(eng.trace(0).copy())
    assert len(traces[0]) == 1500 and np.array_equal(traces[0], traces[1])
    check_sampled_searches(img, mask, p, traces[0], [1499])


def test_dataset_images_all_100():
    """BASELINE configs[0]: the reference's 100 dataset images (256x256, decoded with Pillow), mask
    built on the GPU by the recipe of src/utils.py:79-100 (no perspective stage: it needs OpenCV),
    patch 9, as ONE lock-step batch -- output image, confidence map and per-iteration trace of
    every image against the committed results of the C oracle."""
    import json
    import zlib
    from helpers import GOLDEN
    src = os.path.join(GOLDEN, "xray_images")
    with open(os.path.join(GOLDEN, "xray_dataset_expected.json")) as f:
        want = json.load(f)
    with open(os.path.join(src, "MANIFEST.json")) as f:
        manifest = json.load(f)
    names = sorted(want)
    assert len(names) == 100
    # the pixels the fixtures were made from are committed (decoded_bgr.npz beside the JPEGs: these as Pillow
    # decoded them when make_dataset_expected.py ran), so the test does not depend on this host's decoder
    z = np.load(os.path.join(src, "decoded_bgr.npz"))
    assert [str(n) for n in z["names"]] == names
    imgs = z["bgr"]
    for i, n in enumerate(names):
        assert zlib.crc32(imgs[i].tobytes()) == manifest[n]["crc32_bgr"], n
    with capi.Engine(256, 256, 9, batch=100) as eng:
        eng.upload_automask(imgs, dilation_size=4)
        it, rem = eng.run()
        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        traces = [eng.trace(i) for i in range(100)]
    assert rem == 0 and not msk.any()
    for i, n in enumerate(names):
        w = want[n]
        assert len(traces[i]) == w["iterations"], n
        assert zlib.crc32(np.ascontiguousarray(traces[i]).tobytes()) == w["crc32_trace"], n
        assert zlib.crc32(out[i].tobytes()) == w["crc32_out"], n
        assert zlib.crc32(conf[i].tobytes()) == w["crc32_conf"], n


def test_device_mask_recipe_and_inpaint_image():
    """The GPU mask builder against the NumPy restatement of src/utils.py:79-100, and the
    inpaint_image mirror end to end against the committed fixture of a reference dataset image."""
    from helpers import pipeline_mask
    from crim_hip_inpainter import inpaint_image
    g = load_golden("xray_im001")
    img = g["image"]
    for k in (3, 4, 5):
        with capi.Engine(img.shape[0], img.shape[1], 9) as eng:
            eng.upload_automask(img, dilation_size=k)
            _, _, msk = eng.download(want_mask=True)
        assert np.array_equal(msk[0], pipeline_mask(img, dilation_size=k)), k
    rng = np.random.default_rng(5)
    small = rng.integers(0, 40, (50, 61, 3), dtype=np.uint8)
    with capi.Engine(50, 61, 9) as eng:
        eng.upload_automask(small, dilation_size=4)
        _, _, msk = eng.download(want_mask=True)
    assert np.array_equal(msk[0], pipeline_mask(small, dilation_size=4))
    assert np.array_equal(g["mask"], pipeline_mask(img, dilation_size=4))
    assert np.array_equal(inpaint_image(img, patch_size=9, dilation_size=4), g["out"])


def test_directory_driver_end_to_end(tmp_path):
    """scripts/main.py:36-60 loop shape: real JPEG files in, same-named JPEG files out; what is
    encoded is, image for image, the oracle's result on the decoded pixels with the mask recipe of
    src/utils.py:79-100."""
    from PIL import Image
    from helpers import pipeline_mask
    import crim_batch_driver as drv
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    rng = np.random.default_rng(77)
    shapes = [(64, 80), (64, 80), (64, 80), (72, 72), (64, 80)]
    for i, (h, w) in enumerate(shapes):
        img = crim_synth.xray_like(h, w, 900 + i, colour=(i % 2 == 0))
        img = np.maximum(img, 40)                      # nothing dark by accident
        y0, x0 = int(rng.integers(14, h - 30)), int(rng.integers(14, w - 30))
        img[y0:y0 + 12, x0:x0 + 9] = 0                 # the missing region
        Image.fromarray(img[:, :, ::-1]).save(src / ("im%03d.jpg" % i), quality=98)
    (src / "readme.txt").write_text("not an image")
    (src / "bad.jpg").write_bytes(b"not a jpeg")
    captured = {}

    def writer(path, bgr):
        captured[os.path.basename(path)] = bgr.copy()
        drv.write_bgr(path, bgr)

    msgs = []
    st = drv.inpaint_directory(str(src), str(dst), patch_size=9, dilation_size=4, batch=2,
                               io_threads=3, log=msgs.append, writer=writer)
    assert st["images"] == 5 and st["failed"] == 1 and msgs == ["bad.jpg failed to load."]
    assert sorted(os.listdir(dst)) == ["im%03d.jpg" % i for i in range(5)]
    for i in range(5):
        name = "im%03d.jpg" % i
        dec = drv.read_bgr(str(src / name))
        m = pipeline_mask(dec, dilation_size=4)
        assert m.sum() > 0
        oout, _, _, _, _ = c_oracle.run(dec, m, 9, want_trace=False)
        assert np.array_equal(captured[name], oout), name
        assert drv.read_bgr(str(dst / name)).shape == dec.shape


@pytest.mark.parametrize("nparts", [2, 3, 8])
def test_split_search_virtual_parts(nparts):
    """The window-origin rows divided into parts, each part reduced to one (cost, index) record:
    the result must equal the unsplit search (and the oracle) bit for bit."""
    img = crim_synth.xray_like(160, 144, 1200 + nparts, colour=True)
    m = crim_synth.blob_mask(160, 144, 1200 + nparts, fraction=0.08, margin=9)
    m[0:5, 0:7] = 1      # a border-clipped patch on the way
    oout, oconf, _, otr, n = c_oracle.run(img, m, 9)
    with capi.Engine(160, 144, 9) as eng:
        eng.split_virtual(nparts)
        eng.upload(img, m)
        eng.run()
        out, conf, _ = eng.download(want_conf=True)
        tr = eng.trace(0)
        captured, _ = eng.split_status()
    assert np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE))
    assert np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf)
    assert captured, "the split iterations were not replayed from captured graphs"


@pytest.mark.parametrize("p,eager", [(9, True), (6, False), (15, False)])
def test_split_search_other_loops(p, eager):
    """The split loop launched eagerly, with a patch size that has no select-in-search instance (priority,
    select and update as launches of their own), and at patch edge 15."""
    img = crim_synth.xray_like(120, 136, 1300 + p, colour=True)
    m = crim_synth.blob_mask(120, 136, 1300 + p, fraction=0.07, margin=9)
    oout, oconf, _, otr, n = c_oracle.run(img, m, p)
    with capi.Engine(120, 136, p, flags=capi.FLAG_EAGER if eager else 0) as eng:
        eng.split_virtual(3)
        eng.upload(img, m)
        eng.run()
        out, conf, _ = eng.download(want_conf=True)
        tr = eng.trace(0)
    assert np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE))
    assert np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf)


_RCCL_SCRIPT = r"""
import sys
sys.path[:0] = [%r, %r]
import numpy as np
import crim_hip_capi as capi, crim_synth
from oracle import c_oracle
img = crim_synth.xray_like(96, 96, 1300, colour=True)
m = crim_synth.blob_mask(96, 96, 1300, fraction=0.06, margin=9)
oout, oconf, _, otr, n = c_oracle.run(img, m, 9)
uid = capi.split_unique_id()
assert len(uid) == 128
with capi.Engine(96, 96, 9) as eng:
    eng.split_init(1, 0, uid)
    eng.upload(img, m)
    eng.run()
    out, conf, _ = eng.download(want_conf=True)
    tr = eng.trace(0)
    eng_status = eng.split_status()[0]
assert np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE))
assert np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf)
print("RCCL_SINGLE_RANK_OK", n, "graph captured:", eng_status)
"""


def test_split_search_rccl_single_rank():
    """The RCCL transport with a communicator of one rank (all this box can host): the in-place
    all-gather of the 16-byte records runs every iteration and the result stays exact.  Runs in
    a process of its own: RCCL must find the same ROCm runtime the engine uses, which a process
    that has also imported PyTorch's bundled copy does not guarantee."""
    import subprocess
    import sys
    from helpers import ROOT, PKG
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT % (ROOT, PKG)], capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_split_search_two_processes_on_one_gpu(tmp_path):
    """The split search with MORE than one rank: two worker processes, each with its own replica of a
    1024x1024 image on this one GPU, each scoring only its half of the window-origin rows and consuming the
    other's record every iteration (records exchanged through host memory: RCCL refuses two ranks on one
    device).  Both ranks' per-iteration records must equal the unsplit engine's over 232 iterations, and
    the oracle's over the first ones."""
    import crim_split_worker as sw
    rdv = str(tmp_path / "rdv")
    S, iters = 1024, 200
    procs = [sw.launch_worker(r, 2, 0, rdv, size=S, patch=9, iters=iters, transport="host", dump_trace=True, wait=120)
             for r in range(2)]
    try:
        sw.release(rdv)
        res = sw.collect(rdv, 2, 240.0)
        for pr in procs:
            pr.wait(timeout=60)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    logs = "".join(open(os.path.join(rdv, "worker.%d.log" % r)).read()[-1500:] for r in range(2))
    assert all(r is not None and not r.get("error") for r in res), (res, logs)
    for r in res:
        assert r["transport"] == "host" and r["host_exchanges"] == iters + 32 and not r["graph_captured"]
    assert res[0]["split_matches_single_gpu"]
    traces = [np.load(os.path.join(rdv, "trace.%d.npy" % r)) for r in range(2)]
    img = crim_synth.xray_like(S, S, 4000 + S, colour=True)
    mask = crim_synth.square_hole_mask(S, S, S // 2)
    with capi.Engine(S, S, 9) as eng:
        eng.upload(img, mask)
        eng.run(max_iters=iters + 32)
        want = eng.trace(0)
    assert len(want) == iters + 32
    for r in range(2):
        assert np.array_equal(traces[r], want), "rank %d" % r
    otr = c_oracle.run(img, mask, 9, max_iters=4)[3]
    assert np.array_equal(want[:4], otr.astype(capi.ITER_REC_DTYPE))
