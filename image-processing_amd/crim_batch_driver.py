"""Directory driver for the inpainting stage: the loop shape of /root/reference/scripts/main.py:36-60
(list the ``*.jpg`` files of a directory, load each as BGR uint8, process, write the result under
the same file name) around the reference's ``inpaint_image`` (src/utils.py:67-103), with the
images of one shape filled as lock-step batches on the GPU and JPEG decode / encode running on
host worker PROCESSES while the GPU works (Pillow holds the interpreter lock for part of a decode or
encode, so threads do not scale; the workers are forked before this function first touches the GPU and
never touch it themselves).

Only the inpainting stage is run (the reference's ``process_image`` also fixes perspective,
denoises and adjusts contrast; those stages are not part of this package).  Files are read and
written with Pillow: ``Image.open(...).convert('RGB')`` reversed to B,G,R stands in for
``cv2.imread(path, cv2.IMREAD_COLOR)``, ``save(..., quality=95)`` for ``cv2.imwrite`` (95 is
OpenCV's default JPEG quality).  Decoded pixels can differ from OpenCV's by the usual +-1 of
different JPEG decoders.
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np

import crim_hip_capi as capi


def read_bgr(path):
    """uint8 [H,W,3] in B,G,R order, or None when the file does not decode (cv2.imread returns
    None in that case, scripts/main.py:53-60)."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert('RGB'), dtype=np.uint8)
        return np.ascontiguousarray(rgb[:, :, ::-1])
    except Exception:   # any decode failure is "failed to load", as in the reference
        return None


def write_bgr(path, bgr):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path, quality=95)


def list_tags(path_to_images, start_from=None):
    """File names the reference would process, in its order (scripts/main.py:40-50): every name
    ending in '.jpg' as os.listdir returns them; with start_from set, names before it are skipped
    and processing starts at it."""
    tags = [f for f in os.listdir(path_to_images) if f.endswith('.jpg')]
    if start_from is not None:
        if start_from in tags:
            tags = tags[tags.index(start_from):]
        else:
            tags = []
    return tags


class _ShapeQueue:
    """Images of one shape waiting for a full batch."""

    def __init__(self, shape, batch, make_engine):
        self.shape = shape
        self.batch = batch
        self.engine = make_engine(shape[0], shape[1], batch)
        self.tags = []
        self.images = []

    def flush(self, patch_size, dilation_size):
        """Inpaints the queued images (padding the batch with images that have nothing to fill)
        and returns [(tag, result)]."""
        n = len(self.tags)
        if n == 0:
            return []
        eng = self.engine
        H, W = self.shape
        eng.upload_automask(np.stack(self.images), first=0, dilation_size=dilation_size)
        if n < self.batch:
            pad = self.batch - n
            eng.upload(np.zeros((pad, H, W, 3), np.uint8), np.zeros((pad, H, W), np.uint8), first=n)
        eng.run()
        out, _, _ = eng.download(0, n)
        res = [(t, out[i].copy()) for i, t in enumerate(self.tags)]
        self.tags, self.images = [], []
        return res


def _io_worker_ready(_):
    return os.getpid()


def default_io_workers():
    """Worker processes for JPEG coding: the cores this process may use, at most 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 1:
        n = min(n, env)
    return max(1, min(16, n))


def inpaint_directory(path_to_images, results_dir, patch_size=9, dilation_size=4, batch=256, device=0,
                      io_threads=None, start_from=None, log=None, make_engine=None, reader=read_bgr,
                      writer=write_bgr, io_processes=None):
    """Inpaints every '*.jpg' of `path_to_images` into `results_dir` under the same name.

    Decoding runs ahead on `io_processes` worker processes (default: one per usable core, at most 16),
    encoding and writing trail behind on the same pool, and the calling thread only feeds the GPU.  With
    io_processes=0, or a `reader` / `writer` other than this module's (tests pass closures, which cannot
    cross a process boundary), `io_threads` host threads do the coding instead.  A batch smaller than
    `batch` images of one shape is padded with images that have nothing to fill.  Returns a dict with the
    counts and the wall time.  `make_engine(H, W, batch)`, `reader` and `writer` exist for tests.
    """
    log = log or (lambda s: print(s, file=sys.stderr))
    if make_engine is None:
        def make_engine(H, W, b):
            return capi.Engine(H, W, patch_size, batch=b, device=device)
    os.makedirs(results_dir, exist_ok=True)
    tags = list_tags(path_to_images, start_from)
    t0 = time.time()
    done = failed = 0
    queues = {}
    writes = []
    if io_processes is None:
        io_processes = default_io_workers() if (reader is read_bgr and writer is write_bgr and io_threads is None) else 0
    if io_processes > 0 and not (reader is read_bgr and writer is write_bgr):
        raise ValueError("inpaint_directory: a custom reader or writer needs io_processes=0 (threads)")
    if io_threads is None:
        io_threads = 8
    if io_processes > 0:
        import multiprocessing as mp
        pool_cm = ProcessPoolExecutor(max_workers=io_processes, mp_context=mp.get_context("fork"))
        nworkers = io_processes
    else:
        pool_cm = ThreadPoolExecutor(max_workers=max(1, io_threads))
        nworkers = max(1, io_threads)
    io_threads = nworkers
    with pool_cm as pool:
        if io_processes > 0:
            # every worker exists before the first engine is created (a process forked later would be
            # forked from one that holds a GPU context)
            list(pool.map(_io_worker_ready, range(4 * nworkers)))

        def emit(results):
            for tag, img in results:
                writes.append(pool.submit(writer, os.path.join(results_dir, tag), img))

        # decode at most a few batches ahead of the GPU so that memory stays bounded
        window = max(2 * batch, 4 * io_threads)
        pending = []
        it = iter(tags)
        exhausted = False
        while pending or not exhausted:
            while not exhausted and len(pending) < window:
                try:
                    tag = next(it)
                except StopIteration:
                    exhausted = True
                    break
                pending.append((tag, pool.submit(reader, os.path.join(path_to_images, tag))))
            if not pending:
                break
            tag, fut = pending.pop(0)
            img = fut.result()
            if img is None or img.ndim != 3 or img.shape[2] != 3:
                log(tag + " failed to load.")
                failed += 1
                continue
            shape = img.shape[:2]
            q = queues.get(shape)
            if q is None:
                q = queues[shape] = _ShapeQueue(shape, batch, make_engine)
            q.tags.append(tag)
            q.images.append(img)
            if len(q.tags) == q.batch:
                res = q.flush(patch_size, dilation_size)
                done += len(res)
                emit(res)
        for q in queues.values():
            res = q.flush(patch_size, dilation_size)
            done += len(res)
            emit(res)
            q.engine.close()
        for w in writes:
            w.result()
    return {"images": done, "failed": failed, "seconds": time.time() - t0,
            "shapes": sorted(queues.keys())}


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Inpaint every .jpg of a directory on an MI355X")
    ap.add_argument("path_to_images")
    ap.add_argument("results_dir")
    ap.add_argument("--patch-size", type=int, default=9)
    ap.add_argument("--dilation-size", type=int, default=4)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--io-processes", type=int, default=None, help="JPEG coding workers (default: one per core, at most 16)")
    ap.add_argument("--start-from", default=None)
    a = ap.parse_args(argv)
    st = inpaint_directory(a.path_to_images, a.results_dir, a.patch_size, a.dilation_size, a.batch,
                           a.device, start_from=a.start_from, io_processes=a.io_processes)
    print("%d images in %.2f s (%.1f images/s), %d failed to load" % (
        st["images"], st["seconds"], st["images"] / max(st["seconds"], 1e-9), st["failed"]))


if __name__ == "__main__":
    main()
