"""Expected results for the 100 dataset images under tests/golden/xray_images/ (inputs of the
reference: data, decoded with Pillow; mask by the recipe of /root/reference/src/utils.py:79-100
restated in NumPy, without the perspective stage, which needs OpenCV).

SELF-GOLDEN: the expected values come from this project's C oracle (the reference cannot be run
here); per image: iteration count and CRC-32 of the output image, the final confidence map and the
per-iteration trace records.

    python tests/golden/make_dataset_expected.py
"""
import json
import multiprocessing as mp
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

SRC = os.path.join(ROOT, "tests", "golden", "xray_images")


def one(name):
    import crim_batch_driver as drv
    from helpers import pipeline_mask
    from oracle import c_oracle
    bgr = drv.read_bgr(os.path.join(SRC, name))
    m = pipeline_mask(bgr, dilation_size=4)
    out, conf, _, tr, n = c_oracle.run(bgr, m, 9)
    return name, {"iterations": n, "mask_pixels": int(m.sum()),
                  "crc32_out": zlib.crc32(out.tobytes()),
                  "crc32_conf": zlib.crc32(conf.tobytes()),
                  "crc32_trace": zlib.crc32(np.ascontiguousarray(tr).tobytes())}


def main():
    names = sorted(f for f in os.listdir(SRC) if f.endswith(".jpg"))
    with mp.get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
        res = dict(pool.map(one, names))
    with open(os.path.join(ROOT, "tests", "golden", "xray_dataset_expected.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    # the decoded pixels themselves, so that tests and the benchmark's output check do not depend on the
    # JPEG decoder of the machine they run on
    import crim_batch_driver as drv
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "xray_images", "decoded_bgr.npz"), names=np.array(names),
                        bgr=np.stack([drv.read_bgr(os.path.join(SRC, n)) for n in names]))
    print(len(res), "images,", sum(r["iterations"] for r in res.values()), "iterations")


if __name__ == "__main__":
    main()
