"""The allocation-free one-call entry point (crim_inpaint_noalloc) against the existing entry
points (crim_create + crim_upload_device + crim_run + crim_download_device) on the same device
buffers: identical output bytes and iteration counts, on two calls in a row, the second with a
smaller image (the static state array still holds the first call's contents)."""
import numpy as np
import pytest

import crim_hip_capi as capi
import crim_synth

pytestmark = pytest.mark.gpu


def _case(H, W, seed, colour, p):
    img = crim_synth.xray_like(H, W, seed, colour=colour)
    mask = crim_synth.blob_mask(H, W, seed, fraction=0.07, margin=p)
    return img, mask


def _engine_path(dev, img, mask, p):
    H, W = mask.shape
    d_img, d_mask, d_out = dev.to_device(img), dev.to_device(mask), dev.zeros(img.nbytes)
    with capi.Engine(H, W, p) as eng:
        eng.upload_device(d_img, d_mask)
        eng.run()
        eng.download_device(d_out)
        iters = eng.iterations(0)
    return dev.to_host(d_out, img.shape), iters


def _noalloc_path(dev, img, mask, p):
    H, W = mask.shape
    d_img, d_mask, d_out = dev.to_device(img), dev.to_device(mask), dev.zeros(img.nbytes)
    iters = capi.inpaint_noalloc(H, W, p, d_img, d_mask, d_out)
    return dev.to_host(d_out, img.shape), iters


@pytest.fixture
def dev():
    from helpers import HipBuffers
    capi.load_library()          # brings in the HIP runtime the buffers are allocated with
    d = HipBuffers()
    yield d
    d.free_all()


def test_noalloc_matches_engine_on_two_calls_in_a_row(dev):
    # first a larger colour image, then a smaller grey one with another patch size, then the
    # first again: nothing of an earlier call may leak into a later one
    cases = [(_case(200, 232, 11, True, 9), 9), (_case(72, 96, 12, False, 7), 7),
             (_case(200, 232, 11, True, 9), 9), (_case(40, 33, 13, True, 4), 4)]
    for (img, mask), p in cases:
        want, want_iters = _engine_path(dev, img, mask, p)
        got, got_iters = _noalloc_path(dev, img, mask, p)
        assert got_iters == want_iters
        assert np.array_equal(got, want)
        assert np.array_equal(got[mask == 0], img[mask == 0])


def test_noalloc_largest_shape_and_empty_mask(dev):
    img, mask = _case(512, 512, 14, True, 9)
    mask[:] = 0
    mask[100:108, 200:230] = 1
    want, want_iters = _engine_path(dev, img, mask, 9)
    got, got_iters = _noalloc_path(dev, img, mask, 9)
    assert got_iters == want_iters and np.array_equal(got, want)
    mask[:] = 0
    got, got_iters = _noalloc_path(dev, img, mask, 9)
    assert got_iters == 1 and np.array_equal(got, img)


def test_noalloc_rejects_what_does_not_fit(dev):
    d = dev.zeros(16)
    for H, W in ((513, 512), (1024, 1024), (262144, 1)):
        with pytest.raises(capi.CriminisiHipError) as ei:
            capi.inpaint_noalloc(H, W, 9, d, d, d)
        assert ei.value.code == capi.CRIM_ERR_ARG
    with pytest.raises(capi.CriminisiHipError):
        capi.inpaint_noalloc(64, 64, 0, d, d, d)
    with pytest.raises(capi.CriminisiHipError):
        capi.inpaint_noalloc(64, 64, 9, 0, d, d)
