"""The reference's 100 dataset images as one lock-step batch (BASELINE configs[0]), mask built on the GPU:
prints the time per lock-step iteration.  python3 tools/dataset_run.py [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import crim_hip_capi as capi
import crim_batch_driver as drv
src = os.path.join(ROOT, "tests", "golden", "xray_images")
names = sorted(f for f in os.listdir(src) if f.endswith(".jpg"))
imgs = np.stack([drv.read_bgr(os.path.join(src, n)) for n in names])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with capi.Engine(256, 256, 9, batch=len(names), flags=capi.FLAG_NO_TRACE) as eng:
    eng.upload_automask(imgs, dilation_size=4)
    eng.run()
    t = time.time()
    for _ in range(reps):
        eng.upload_automask(imgs, dilation_size=4)
        it, rem = eng.run()
    dt = (time.time() - t) / reps
    print("%d images, %d lock-step iterations in %.4f s: %.1f us per lock-step iteration, %.0f images/s"
          % (len(names), it, dt, 1e6 * dt / it, len(names) / dt))
