#!/usr/bin/env python3
"""Extended random parity sweep (GPU engine against the C oracle), larger and more varied than
the one in the test suite: image sides 20..420 (several search tiles per row and column), every
patch edge 3..17, grey / colour / noise / flat images, masks with border contact and salt, and
the many-row-tiles-per-block walk of the pipelined search forced on or off, and the iteration run as two, three
or four launches.

    python tools/stress_sweep.py [--cases 200] [--seed 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]

import numpy as np  # noqa: E402
import crim_hip_capi as capi  # noqa: E402
import crim_synth  # noqa: E402
from oracle import c_oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = 0
    t0 = time.time()
    for k in range(a.cases):
        H = int(rng.integers(20, 420 if k % 3 == 0 else 120))
        W = int(rng.integers(20, 420 if k % 3 == 1 else 120))
        p = int(rng.choice([3, 4, 5, 6, 7, 8, 9, 9, 9, 11, 13, 15, 17]))
        kind = k % 4
        if kind == 0:
            img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == 1:
            img = crim_synth.xray_like(H, W, 19000 + k, colour=True)
        elif kind == 2:
            img = crim_synth.xray_like(H, W, 19000 + k, colour=False)
        else:
            img = np.full((H, W, 3), int(rng.integers(0, 256)), np.uint8)
            img[:, W // 2:] = int(rng.integers(0, 256))
        m = np.zeros((H, W), np.uint8)
        for _ in range(int(rng.integers(1, 5))):
            h = int(rng.integers(1, max(2, min(H // 3, 40))))
            w = int(rng.integers(1, max(2, min(W // 3, 40))))
            y0 = int(rng.integers(0, H - h + 1))
            x0 = int(rng.integers(0, W - w + 1))
            m[y0:y0 + h, x0:x0 + w] = 1
        if k % 5 == 0:
            m[rng.random((H, W)) < 0.01] = 1
        if m.all():
            m[0, 0] = 0
        os.environ["CRIM_PIPE_MIN_BLOCKS"] = "1" if k % 2 else "1024"
        os.environ["CRIM_PIPE_T"] = str(int(rng.choice([1, 2, 3, 8])))
        # the iteration as two launches (default for a single image), three (CRIM_FUSED=0) or four (the
        # selection in its own launch, as large batches run)
        mode = int(rng.integers(0, 3))
        for name in ("CRIM_FUSED", "CRIM_SELECT_IN_SEARCH"):
            os.environ.pop(name, None)
        if mode == 1:
            os.environ["CRIM_FUSED"] = "0"
        elif mode == 2:
            os.environ["CRIM_SELECT_IN_SEARCH"] = "0"
        oout, oconf, _, otr, n = c_oracle.run(img, m, p)
        with capi.Engine(H, W, p) as eng:
            eng.upload(img, m)
            try:
                eng.run()
            except capi.CriminisiHipError as e:
                # an input whose fill makes no progress is reported by both sides in their own way
                print("case %d (H=%d W=%d p=%d kind=%d): engine error %s" % (k, H, W, p, kind, e))
                bad += 1
                continue
            out, conf, _ = eng.download(want_conf=True)
            tr = eng.trace(0)
        ok = (len(tr) == n and np.array_equal(out[0], oout) and np.array_equal(conf[0], oconf)
              and np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE)))
        if not ok:
            bad += 1
            print("case %d (H=%d W=%d p=%d kind=%d T=%s min=%s launches=%d): MISMATCH" % (
                k, H, W, p, kind, os.environ["CRIM_PIPE_T"], os.environ["CRIM_PIPE_MIN_BLOCKS"], 2 + mode), flush=True)
        if k % 20 == 19:
            print("%d cases, %d bad, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d bad" % (a.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
