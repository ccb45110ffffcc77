"""Time of the noise-filter stage on 100 synthetic 256x256 images (host arrays in and out).
python3 tools/nlm_time.py   (CRIM_NLM_DIRECT=1: the window-difference kernels; CRIM_NLM_CORR=1: correlation form without shared row products)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import crim_hip_capi as capi, crim_synth
if os.environ.get("NLM_SYNTH"):
    imgs = np.stack([crim_synth.xray_like(256, 256, 9000 + i, colour=True) for i in range(100)])
else:   # the reference's 100 dataset images (decoded pixels committed beside the JPEGs)
    imgs = np.load(os.path.join(ROOT, "tests", "golden", "xray_images", "decoded_bgr.npz"))["bgr"]
capi.filter_noise(imgs[:4])
t = time.time(); out = capi.filter_noise(imgs, median_ksize=3, nlm_h=7, apply_sharpening=True); dt = time.time() - t
import zlib
print("%s: 100 images in %.1f ms: %.0f images/s, crc %08x" % ("differences" if os.environ.get("CRIM_NLM_DIRECT") else "correlation" if os.environ.get("CRIM_NLM_CORR") else "shared rows", 1e3 * dt, 100 / dt, zlib.crc32(out.tobytes())))
