"""One lane of the headline workload (4096 grey 256x256 images, 10 % masks, patch 9) run to completion on
one stream: under rocprofv3 --kernel-trace --stats the kernel averages are those of each launch alone
on the GPU.  python3 tools/solo_profile.py [batch [max iterations [size [patch]]]]
(size > 256: one false-colour image with a centred square hole of half its edge, as BASELINE configs[3] / [4])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import crim_hip_capi as capi
import crim_synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
PATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 9
if H > 256:
    img = crim_synth.xray_like(H, W, 4000 + H, colour=True)
    mask = crim_synth.square_hole_mask(H, W, H // 2)
    with capi.Engine(H, W, PATCH, flags=capi.FLAG_NO_TRACE) as eng:
        eng.upload(img, mask)
        t = time.time()
        it, rem = eng.run(max_iters=int(sys.argv[2]))
        dt = time.time() - t
        print("%d x %d, patch %d: %d iterations in %.3f s: %.1f us per iteration" % (H, W, PATCH, it, dt, 1e6 * dt / it))
    sys.exit(0)
base = [(crim_synth.xray_like(H, W, 1000 + i), crim_synth.blob_mask(H, W, 1000 + i, 0.10, 9)) for i in range(64)]
imgs = np.stack([base[i % 64][0] for i in range(B)])
masks = np.stack([base[i % 64][1] for i in range(B)])
with capi.Engine(H, W, 9, batch=B, flags=capi.FLAG_ONE_LANE | capi.FLAG_NO_TRACE) as eng:
    eng.upload(imgs, masks)
    t = time.time()
    it, rem = eng.run(max_iters=int(sys.argv[2])) if len(sys.argv) > 2 else eng.run()
    dt = time.time() - t
    print("%d images, %d lock-step iterations in %.3f s: %.0f images/s, %.1f us per iteration" % (B, it, dt, B / dt, 1e6 * dt / it))
