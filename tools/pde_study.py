#!/usr/bin/env python3
"""Offline study: how much of the search's multiply-accumulate work partial-distortion
elimination would skip.  Runs the NumPy oracle for some iterations and, per iteration, simulates
(1) a bound taken from a sub-lattice of windows scored completely, (2) wave-granular early exit
(128 x 8 origins) once every window of the wave has a partial SSD above the bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import numpy as np
from oracle import criminisi_np as cn, cv2_restated as cvr
import crim_synth

def study(img, mask, p, iters, every, checks=(3, 5, 7), wave=(8, 128), lat=(4, 1)):
    o = cn.CriminisiOracle(img, mask, p)
    fr = []
    for it in range(iters):
        if o.working_mask.sum() == 0:
            break
        if it % every == 0:
            o._find_front(); o._update_priorities()
            tp = o._find_highest_priority_pixel()
            pt = o._get_patch(tp)
            ph = 1 + pt[0][1] - pt[0][0]; pw = 1 + pt[1][1] - pt[1][0]
            lab = cvr.bgr2lab_u8(o.working_image).astype(np.int64)
            m = o.working_mask.astype(np.int64)
            k = 1 - m[pt[0][0]:pt[0][1] + 1, pt[1][0]:pt[1][1] + 1]
            T = lab[pt[0][0]:pt[0][1] + 1, pt[1][0]:pt[1][1] + 1]
            oh, ow = o.iheight - ph + 1, o.iwidth - pw + 1
            rows = np.zeros((ph, oh, ow), np.int64)
            for r in range(ph):
                for j in range(pw):
                    if k[r, j]:
                        d = lab[r:r + oh, j:j + ow] - T[r, j]
                        rows[r] += (d * d).sum(axis=2)
            cum = np.cumsum(rows, axis=0)
            D = cum[-1]
            inval = cvr.window_sum_topleft(o.working_mask.astype(np.uint8), ph, pw)[:oh, :ow] > 0
            ty = (pt[0][0] + pt[0][1]) / 2.0; tx = (pt[1][0] + pt[1][1]) / 2.0
            dist = np.sqrt((np.arange(oh)[:, None] + ph / 2.0 - ty) ** 2 + (np.arange(ow)[None, :] + pw / 2.0 - tx) ** 2)
            cost = np.where(inval, np.inf, D + dist)
            best = cost.min()
            # bound from the lattice
            lc = cost[::lat[0], ::lat[1]]
            bound = np.floor(lc.min()) + 1
            # wave-granular pruning
            wy, wx = wave
            work = 0.0; total = 0.0
            latfrac = 1.0 / (lat[0] * lat[1])
            for y0 in range(0, oh, wy):
                for x0 in range(0, ow, wx):
                    sub = cum[:, y0:y0 + wy, x0:x0 + wx]
                    done = ph
                    for c in checks:
                        if c < ph and sub[c - 1].min() > bound:
                            done = c
                            break
                    total += ph
                    work += latfrac * ph + (1 - latfrac) * done
            fr.append((it, work / total, best, bound, np.median(D[~inval])))
        o.step()
    return fr

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "synth"
    if which == "synth":
        img = crim_synth.xray_like(256, 256, 7, colour=False)
        mask = crim_synth.blob_mask(256, 256, 7, fraction=0.10, margin=9)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers
        img, mask = helpers.load_golden_dataset(which) if hasattr(helpers, "load_golden_dataset") else (None, None)
    for lat in ((4, 1), (4, 4)):
        fr = study(img, mask, 9, 300, 20, lat=lat)
        print("lattice", lat)
        for f in fr:
            print("  it %4d work %.3f best %.1f bound %.1f medianD %.0f" % f)
        print("  mean work fraction %.3f" % np.mean([f[1] for f in fr]))
