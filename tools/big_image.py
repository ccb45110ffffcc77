"""Large-image check (BASELINE configs[3]): 2048x2048x3, 1024x1024 hole, patch 9 / 15.
Compares the first iterations with the C oracle, then times the engine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import crim_hip_capi as capi, crim_synth
from oracle import c_oracle
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
p = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ncheck = int(sys.argv[3]) if len(sys.argv) > 3 else 10
nrun = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
img = crim_synth.xray_like(S, S, 4000 + S, colour=os.environ.get("BIG_GREY", "0") != "1")   # BIG_GREY=1: grey engine
mask = crim_synth.square_hole_mask(S, S, S // 2)
t = time.time(); oout, oconf, omask, otr, n = c_oracle.run(img, mask, p, max_iters=ncheck); tcpu = time.time() - t
print("oracle: %d iterations in %.1f s (%.3f s/iter)" % (n, tcpu, tcpu / n), flush=True)
with capi.Engine(S, S, p) as eng:
    eng.upload(img, mask)
    eng.run(max_iters=ncheck)
    out, conf, msk = eng.download(want_conf=True, want_mask=True)
    tr = eng.trace(0)
    ok = (np.array_equal(tr, otr.astype(capi.ITER_REC_DTYPE)) and np.array_equal(out[0], oout)
          and np.array_equal(conf[0], oconf) and np.array_equal(msk[0], omask))
    print("first %d iterations bit-exact vs oracle:" % ncheck, ok, flush=True)
    if not ok:
        for i in range(len(tr)):
            if tr[i] != otr.astype(capi.ITER_REC_DTYPE)[i]:
                print("  first diff at", i, tr[i], otr[i]); break
    t = time.time(); it, rem = eng.run(max_iters=nrun); dt = time.time() - t
    P = 2 * ((p - 1) // 2) + 1
    nc = (S - P + 1) ** 2
    print("engine: %d more iterations in %.3f s -> %.1f us/iter, %.3g evals/s, %.2f TMAC/s (whole iteration)"
          % (nrun, dt, 1e6 * dt / nrun, nc * nrun / dt, nc * nrun / dt * P * P * 3 / 1e12), flush=True)
    if len(sys.argv) > 5:
        t = time.time(); it, rem = eng.run(); dt2 = time.time() - t
        out, _, msk = eng.download(want_mask=True)
        print("full run: %d iterations total, remaining %d, %.2f s for the rest; known pixels untouched: %s"
              % (it, rem, dt2, np.array_equal(out[0][mask == 0], img[mask == 0])), flush=True)
