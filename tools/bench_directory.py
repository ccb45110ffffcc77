#!/usr/bin/env python3
"""Throughput of the directory driver (crim_batch_driver.inpaint_directory, the loop shape of
/root/reference/scripts/main.py:36-60): N synthetic 256x256 JPEGs with near-black holes (the
damage the reference's mask recipe, src/utils.py:79-100, looks for) are written to a scratch
directory, inpainted into another, and the wall time of the driver call is reported.

    python tools/bench_directory.py [--images 2048] [--batch 512] [--io-processes 16] [--dataset]

--dataset: the reference's own 100 dataset JPEGs (tests/golden/xray_images), as scripts/main.py walks them.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]

import numpy as np  # noqa: E402
import crim_batch_driver as drv  # noqa: E402
import crim_synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--io-processes", type=int, default=None)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dataset", action="store_true")
    a = ap.parse_args()
    S = a.size
    if a.dataset:
        src = os.path.join(ROOT, "tests", "golden", "xray_images")
        dst = tempfile.mkdtemp(prefix="crim_out_")
        try:
            st = drv.inpaint_directory(src, dst, patch_size=9, dilation_size=4, batch=100, io_processes=a.io_processes,
                                       log=lambda s: None)
            n_out = len([f for f in os.listdir(dst) if f.endswith(".jpg")])
            print(json.dumps({"images": st["images"], "failed": st["failed"], "written": n_out, "seconds": st["seconds"],
                              "images_per_s": st["images"] / st["seconds"], "batch": 100,
                              "io_processes": a.io_processes or drv.default_io_workers(), "size": 256, "dataset": True}))
        finally:
            shutil.rmtree(dst, ignore_errors=True)
        return
    src = tempfile.mkdtemp(prefix="crim_in_")
    dst = tempfile.mkdtemp(prefix="crim_out_")
    try:
        t0 = time.time()
        # 64 distinct images, written under a.images names (synthesis costs more than the run measured here)
        blobs = []
        for i in range(min(64, a.images)):
            img = crim_synth.xray_like(S, S, 7000 + i, colour=False)
            img = np.maximum(img, 40)                       # nothing but the holes is near-black
            hole = crim_synth.blob_mask(S, S, 7000 + i, fraction=0.04, margin=24) != 0
            img[hole] = 0
            path = os.path.join(src, "im%05d.jpg" % i)
            drv.write_bgr(path, img)
            with open(path, "rb") as f:
                blobs.append(f.read())
        for i in range(len(blobs), a.images):
            with open(os.path.join(src, "im%05d.jpg" % i), "wb") as f:
                f.write(blobs[i % len(blobs)])
        t_make = time.time() - t0
        st = drv.inpaint_directory(src, dst, patch_size=9, dilation_size=4, batch=a.batch,
                                   io_processes=a.io_processes, log=lambda s: None)
        n_out = len([f for f in os.listdir(dst) if f.endswith(".jpg")])
        print(json.dumps({"images": st["images"], "failed": st["failed"], "written": n_out,
                          "seconds": st["seconds"], "images_per_s": st["images"] / st["seconds"],
                          "batch": a.batch, "io_processes": a.io_processes or drv.default_io_workers(), "size": S,
                          "make_inputs_seconds": t_make}))
    finally:
        shutil.rmtree(src, ignore_errors=True)
        shutil.rmtree(dst, ignore_errors=True)


if __name__ == "__main__":
    main()
