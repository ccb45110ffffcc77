"""Where the time of the streamed (host-I/O-inclusive) batch goes: per-thread wall time of upload,
fill and download with two half-batch engines on two threads."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import numpy as np
import bench
import crim_hip_capi as capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
imgs, masks = bench.make_batch(B, 256, 256, 2, 0, False)
part = B // nthreads
pin_i, pin_m, out = capi.pinned_empty(imgs.shape), capi.pinned_empty(masks.shape), capi.pinned_empty(imgs.shape)
pin_i[:] = imgs; pin_m[:] = masks
engines = [capi.Engine(256, 256, 9, batch=part, flags=capi.FLAG_NO_TRACE | capi.FLAG_ONE_LANE) for _ in range(nthreads)]
log = [[] for _ in range(nthreads)]
link = threading.Lock()   # one transfer phase at a time: the others' kernels run under it
def loop(k, n):
    lo = k * part
    for _ in range(n):
        t0 = time.perf_counter()
        with link:
            engines[k].upload(pin_i[lo:lo + part], pin_m[lo:lo + part])
        t1 = time.perf_counter(); engines[k].run()
        t2 = time.perf_counter()
        with link:
            engines[k].download(out=out[lo:lo + part])
        t3 = time.perf_counter(); log[k].append((t0, t1, t2, t3))
def timed(n):
    ts = [threading.Thread(target=loop, args=(k, n)) for k in range(nthreads)]
    t0 = time.perf_counter()
    [t.start() for t in ts]; [t.join() for t in ts]
    return time.perf_counter() - t0
timed(1)
for l in log: l.clear()
dt = timed(3)
print("images/s", B * 3 / dt, "ms per pass", 1e3 * dt / 3)
base = min(l[0][0] for l in log)
for k, l in enumerate(log):
    for (a, b, c, d) in l:
        print("thread %d: upload %6.1f..%6.1f (%5.1f ms)  run ..%6.1f (%6.1f ms)  download ..%6.1f (%5.1f ms)" %
              (k, 1e3 * (a - base), 1e3 * (b - base), 1e3 * (b - a), 1e3 * (c - base), 1e3 * (c - b), 1e3 * (d - base), 1e3 * (d - c)))
