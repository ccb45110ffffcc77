"""A batch of synthetic 512x512 false-colour images with 10 % masks (BASELINE configs[2], one lane's worth by
default): time per lock-step iteration over the first iterations.  python3 tools/batch512_run.py [batch [iterations]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import crim_hip_capi as capi
import crim_synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 800
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
base = [(crim_synth.xray_like(512, 512, 3000 + i, colour=True), crim_synth.blob_mask(512, 512, 3000 + i, 0.10, 9)) for i in range(16)]
imgs = np.stack([base[i % 16][0] for i in range(B)])
masks = np.stack([base[i % 16][1] for i in range(B)])
with capi.Engine(512, 512, 9, batch=B, flags=capi.FLAG_NO_TRACE) as eng:
    eng.upload(imgs, masks)
    eng.run(max_iters=64)
    t = time.time()
    it, rem = eng.run(max_iters=N)
    dt = time.time() - t
    print("%d images of 512x512: %d lock-step iterations in %.3f s: %.1f us per iteration" % (B, N, dt, 1e6 * dt / N))
