import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import *
import numpy as np
import crim_hip_capi as capi
from oracle import c_oracle
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import rand_case, engine_run
for p in (13, 15, 11):
    img, m = rand_case(100 + p, 72, 88, noise=(p % 4 == 1), colour=True)
    out, conf, msk, tr = engine_run(img, m, p)
    oout, oconf, omask, otr, n = c_oracle.run(img, m, p)
    bad = np.argwhere(conf != oconf)
    print("p", p, "iters", n, len(tr), "bad conf pixels", len(bad))
    for (y, x) in bad[:10]:
        print("  ", y, x, "mask0", m[y, x], conf[y, x], oconf[y, x])
    if len(bad):
        # find iteration at which the pixel was last in a patch box
        for (y, x) in bad[:3]:
            half = (p - 1) // 2
            its = [i for i, r in enumerate(otr) if abs(r["tr"] - y) <= half + 2 and abs(r["tc"] - x) <= half + 2]
            print("   near-target iterations for", y, x, its[-5:], "of", n)
