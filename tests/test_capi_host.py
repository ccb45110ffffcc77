"""CPU tests of the C-ABI library and the host-side mirror: the library loads, exports every
symbol the header declares, builds the same Lab tables as the oracle, and reports errors
instead of aborting.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import ROOT

import crim_hip_capi as capi
from oracle import cv2_restated as cvr


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    header = open(os.path.join(ROOT, "include", "criminisi_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(crim_[a-z_]+)\s*\(", header)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libcriminisi_hip.so does not export " + n


def test_lab_tables_match_oracle():
    g, c, k = capi.lab_tables()
    assert np.array_equal(g, cvr.SRGB_GAMMA_TAB)
    assert np.array_equal(c, cvr.LAB_CBRT_TAB)
    assert np.array_equal(k, cvr.LAB_COEFFS.astype(np.int32))


def test_create_rejects_bad_arguments():
    lib = capi.load_library()
    h = ctypes.c_void_p()
    assert lib.crim_create(0, 16, 9, 0, 0, ctypes.byref(h)) == capi.CRIM_ERR_ARG
    assert lib.crim_create(16, 16, 0, 0, 0, ctypes.byref(h)) == capi.CRIM_ERR_ARG
    assert lib.crim_create(16, 16, capi.MAX_PATCH + 1, 0, 0, ctypes.byref(h)) == capi.CRIM_ERR_ARG
    assert lib.crim_create(16, 16, 9, 0, 0, None) == capi.CRIM_ERR_ARG
    assert b"patch_size" in lib.crim_last_error() or b"out is null" in lib.crim_last_error()
    assert lib.crim_run(None, 0, None, None) == capi.CRIM_ERR_ARG
    lib.crim_destroy(None)


def test_record_layout():
    assert capi.ITER_REC_DTYPE.itemsize == 32


def test_shim_validation_error_comes_from_inpaint():
    from src.criminisi_inpainter import Criminisi_Inpainter, CriminisiInpainter
    assert CriminisiInpainter is Criminisi_Inpainter
    img = np.zeros((8, 8, 3), np.uint8)
    mask = np.zeros((8, 9), np.uint8)
    ip = Criminisi_Inpainter(img, mask, patch_size=9)         # no error here (reference :18-38)
    with pytest.raises(AttributeError, match='mask and image must be of the same size'):
        ip.inpaint()                                          # raised at :44 / :90-92


def test_shim_public_attributes():
    from src.criminisi_inpainter import Criminisi_Inpainter
    img = np.arange(8 * 8 * 3, dtype=np.uint8).reshape(8, 8, 3)
    mask = np.zeros((8, 8), np.uint8)
    mask[3, 3] = 1
    ip = Criminisi_Inpainter(img, mask, 9, False, False)
    for name in ("image", "mask", "patch_size", "verbose", "show_progress", "iheight", "iwidth",
                 "working_image", "working_mask", "front", "confidence", "data", "priority"):
        assert hasattr(ip, name)
    assert ip.iheight == 8 and ip.iwidth == 8
    assert ip.confidence[3, 3] == 0.0 and ip.confidence[0, 0] == 1.0
    assert ip.working_image is not img and np.array_equal(ip.working_image, img)


def test_shim_rejects_non_bgr():
    from src.criminisi_inpainter import Criminisi_Inpainter
    ip = Criminisi_Inpainter(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8))
    with pytest.raises(ValueError):
        ip.inpaint()


@pytest.mark.skipif(capi.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_product_fails_loudly_without_gpu():
    from src.criminisi_inpainter import Criminisi_Inpainter
    ip = Criminisi_Inpainter(np.zeros((16, 16, 3), np.uint8), np.zeros((16, 16), np.uint8))
    with pytest.raises(capi.CriminisiHipError):
        ip.inpaint()


def test_product_does_not_import_oracle_or_cv2():
    pkg = os.path.join(ROOT, "image-processing_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn
    src = open(os.path.join(ROOT, "src", "criminisi_inpainter.py")).read()
    assert "import cv2" not in src and "from oracle" not in src


def test_shim_warns_beyond_the_exact_regime_of_the_reference():
    import warnings
    from src.criminisi_inpainter import Criminisi_Inpainter
    img = np.zeros((40, 40, 3), np.uint8)
    mask = np.zeros((40, 40), np.uint8)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Criminisi_Inpainter(img, mask, patch_size=9)
        assert not w
        Criminisi_Inpainter(img, mask, patch_size=17)
        assert len(w) == 1 and "float32" in str(w[0].message)


def test_noalloc_state_fits_its_static_array_for_every_promised_shape():
    """crim_inpaint_noalloc carves all engine state of one image out of a static device array of
    CRIM_NOALLOC_BYTES; the header promises room for every shape of at most 512 * 512 pixels with at least
    32 rows and 32 columns.  The layout is computed on the host, so the promise is checked here for
    the extreme shapes and every patch size (a plane added to the engine state shows up in this test,
    not as a refused call)."""
    lib = capi.load_library()
    worst = 0
    for h, w in ((512, 512), (8192, 32), (32, 8192), (1024, 256), (256, 1024), (511, 513), (64, 4096), (4096, 64)):
        assert h * w <= capi.NOALLOC_MAX_PIXELS
        for patch in (1, 3, 9, 15, 16, 31):
            need = capi.noalloc_bytes_needed(h, w, patch)
            assert 0 < need <= capi.NOALLOC_BYTES, (h, w, patch, need)
            worst = max(worst, need)
    assert worst > 0.8 * capi.NOALLOC_BYTES      # the constant is not a wild over-estimate either
    # a single row of 262144 pixels is refused: the search planes end in 64 spare rows of that width
    assert capi.noalloc_bytes_needed(1, 262144, 9) > capi.NOALLOC_BYTES
    assert capi.noalloc_bytes_needed(0, 5, 9) == 0 and lib.crim_noalloc_bytes_needed(5, 5, 99) == 0
