"""Generates the committed fixtures under tests/golden/ (run once, in the build container).

SELF-GOLDEN: the reference holds no fixtures and cannot be imported here (no OpenCV), so the
expected outputs come from this project's NumPy oracle (oracle/criminisi_np.py).  Inputs of the
two ``xray_*`` cases are dataset images of the reference (data, decoded with Pillow) with the
mask recipe of /root/reference/src/utils.py:79-100 restated in NumPy (no perspective stage).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import crim_synth  # noqa: E402
from helpers import pipeline_mask  # noqa: E402
from oracle.criminisi_np import CriminisiOracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, img, mask, p):
    o = CriminisiOracle(img, mask, p)
    out = o.inpaint()
    tr = np.array([(a, b, c, d) for a, b, c, d, e, f in o.trace], dtype=np.int32).reshape(-1, 4)
    cost = np.array([e for *_, e, f in o.trace], dtype=np.float64)
    rem = np.array([f for *_, f in o.trace], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), image=img, mask=mask, patch_size=p,
                        out=out, confidence=o.confidence, trace=tr, cost=cost, remaining=rem)
    print(name, img.shape, int(mask.sum()), "iters", len(cost))


def main():
    rng = np.random.default_rng(20240607)
    # synthetic: noise, smooth grey, false colour; interior and border-touching masks
    img = rng.integers(0, 256, (48, 56, 3), dtype=np.uint8)
    m = np.zeros((48, 56), np.uint8); m[18:30, 20:34] = 1
    save("noise_p9", img, m, 9)
    img = crim_synth.xray_like(64, 64, 11)
    m = crim_synth.blob_mask(64, 64, 11, fraction=0.10, margin=9)
    save("grey_p9", img, m, 9)
    img = crim_synth.xray_like(60, 72, 12, colour=True)
    m = np.zeros((60, 72), np.uint8); m[0:9, 0:12] = 1; m[25:40, 60:72] = 1; m[52:60, 30:41] = 1
    save("colour_border_p7", img, m, 7)
    img = crim_synth.xray_like(56, 56, 13, colour=True)
    m = crim_synth.blob_mask(56, 56, 13, fraction=0.08, margin=8)
    save("colour_p15", img, m, 15)
    img = crim_synth.xray_like(40, 44, 14)
    m = np.zeros((40, 44), np.uint8); m[10:14, 10:30] = 1; m[20, 20] = 1
    save("grey_p4_even", img, m, 4)
    # dataset images of the reference (data only)
    src = "/root/reference/xray_images"
    if os.path.isdir(src):
        from PIL import Image
        for tag in ("im001-healthy.jpg", "im050-pneumonia.jpg"):
            path = os.path.join(src, tag)
            if not os.path.exists(path):
                cands = sorted(os.listdir(src))
                path = os.path.join(src, cands[len(cands) // 2])
                tag = os.path.basename(path)
            rgb = np.asarray(Image.open(path).convert("RGB"))
            bgr = np.ascontiguousarray(rgb[:, :, ::-1])
            m = pipeline_mask(bgr, dilation_size=4)
            save("xray_" + tag.split("-")[0], bgr, m, 9)


if __name__ == "__main__":
    main()
