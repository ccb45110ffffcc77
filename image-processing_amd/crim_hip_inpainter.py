"""Host-side mirror of the reference's inpainter class, backed by the HIP engine.

Mirrors /root/reference/src/criminisi_inpainter.py: class name, constructor signature (:18),
public attributes (:19-38), ``inpaint()`` (:40-88) with its validation error (:90-92) and its
``verbose`` / ``show_progress`` side effects (:55-57,:67-69,:76-77,:85-87).  All arithmetic runs
on the GPU through libcriminisi_hip.so; there is no CPU path.
"""
import os
import time

import numpy as np

import crim_hip_capi as capi


def _write_png(path, bgr):
    """show_progress frame dump (criminisi_inpainter.py:56-57 uses cv2.imwrite on the BGR
    array); OpenCV is used when the host has it, Pillow otherwise."""
    try:
        import cv2  # noqa: F401  (present on machines that run the full reference pipeline)
        cv2.imwrite(path, bgr)
        return
    except ImportError:
        pass
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)


class _FrameDumper:
    """Rebuilds the working image iteration by iteration from trace records (the update rule of
    criminisi_inpainter.py:267-308: masked pixels of the target patch take the source patch's
    pixels, then the patch's mask is cleared) and writes results/working_image_<c>.png for the
    state BEFORE iteration c, as the reference does at :55-57, on worker threads."""

    def __init__(self, image, mask_u8, patch_size, workers=4):
        from concurrent.futures import ThreadPoolExecutor
        self.img = np.array(image, dtype=np.uint8, copy=True)
        self.msk = np.array(mask_u8, dtype=np.uint8, copy=True)
        self.half = (int(patch_size) - 1) // 2
        self.c = 0
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.pending = []

    def emit_then_apply(self, rec):
        self.pending.append(self.pool.submit(_write_png, 'results/working_image_' + str(self.c) + '.png',
                                             self.img.copy()))
        self.c += 1
        H, W = self.msk.shape
        tr, tc, sy, sx = int(rec['tr']), int(rec['tc']), int(rec['sy']), int(rec['sx'])
        r0, r1 = max(0, tr - self.half), min(tr + self.half, H - 1)
        c0, c1 = max(0, tc - self.half), min(tc + self.half, W - 1)
        ph, pw = r1 - r0 + 1, c1 - c0 + 1
        src = self.img[sy:sy + ph, sx:sx + pw].copy()
        sel = self.msk[r0:r1 + 1, c0:c1 + 1] != 0
        self.img[r0:r1 + 1, c0:c1 + 1][sel] = src[sel]
        self.msk[r0:r1 + 1, c0:c1 + 1] = 0
        if len(self.pending) > 64:          # bound the frames held in memory
            self.pending.pop(0).result()

    def close(self):
        for f in self.pending:
            f.result()
        self.pending = []
        self.pool.shutdown(wait=True)


class Criminisi_Inpainter():
    '''
    Criminisi (2004) exemplar-based inpainting on an AMD MI355X.

    Requires binary mask of missing region, same size as image.
        Pixels values: 1 for missing, 0 for known region.
    '''

    VERBOSE_CHUNK = 64   # iterations between two bursts of verbose output

    def __init__(self, image, mask, patch_size=9, verbose=False, show_progress=False, device=0):
        self.image = image
        self.mask = mask.round()
        self.patch_size = patch_size
        self.verbose = verbose
        self.show_progress = show_progress
        self.device = device

        if patch_size > 16:
            import warnings
            warnings.warn('patch_size %d: beyond 16 the reference\'s float32 sums of squared differences exceed '
                          '2**24 and round; this engine keeps exact integers, so its choices can differ from '
                          'the reference\'s there' % patch_size, stacklevel=2)
        self.iheight, self.iwidth = self.image.shape[:2]
        self.working_image = np.copy(self.image)
        self.working_mask = np.copy(self.mask)
        self.confidence = (1.0 - self.mask)
        self._front = np.zeros([self.iheight, self.iwidth])
        self._data = np.zeros([self.iheight, self.iwidth])
        self._priority = np.zeros([self.iheight, self.iwidth])
        self._engine = None
        self._fields_stale = False
        self.iterations = 0
        self.search_seconds = None   # verbose runs: per-iteration 'Time to find best', in seconds

    # front / data / priority (criminisi_inpainter.py:31,37,38) are materialised on demand: the
    # fill loop never needs them as full frames.  After inpaint() the reference holds the frames its
    # LAST iteration computed before its update (:53-59); they are reproduced, when first read, by
    # filling the same input once more up to the last iteration and evaluating the fields there.
    # Cost: the first read of any of the three after inpaint() runs the whole fill a second time on an
    # engine of its own (as long as inpaint() itself took); later reads are free.  Callers that never
    # read them -- the reference's own pipeline does not -- pay nothing.
    def _refresh_fields(self):
        if self._fields_stale and self._engine is not None:
            img = np.ascontiguousarray(self.image, dtype=np.uint8)
            mask_u8 = (np.asarray(self.mask) != 0).astype(np.uint8)
            with capi.Engine(self.iheight, self.iwidth, self.patch_size, batch=1, device=self.device,
                             flags=capi.FLAG_NO_TRACE) as eng:
                eng.upload(img, mask_u8)
                if self.iterations > 1:
                    eng.run(max_iters=self.iterations - 1)
                f, d, p = eng.peek_fields(0)
            self._front, self._data, self._priority = f, d, p
            self._fields_stale = False

    @property
    def front(self):
        self._refresh_fields()
        return self._front

    @property
    def data(self):
        self._refresh_fields()
        return self._data

    @property
    def priority(self):
        self._refresh_fields()
        return self._priority

    def inpaint(self):
        '''
        Compute the new image and return it.
        '''
        self._validate_inputs()
        img = np.asarray(self.image)
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise ValueError('image must be a uint8 array of shape (H, W, 3) in BGR order')
        mask_u8 = (np.asarray(self.mask) != 0).astype(np.uint8)

        start_time = time.time()
        total = int(np.sum(self.mask))
        eng = capi.Engine(self.iheight, self.iwidth, self.patch_size, batch=1, device=self.device)
        self._engine = eng
        eng.upload(img, mask_u8)

        if self.verbose or self.show_progress:
            # Neither side effect puts the host into the loop: the fill keeps replaying graphs, a
            # chunk of iterations per call, and both are produced behind it from the device's
            # per-iteration records.  verbose (:67-69, :76-77): the lines of a chunk are printed
            # together; 'Time to find best' is that iteration's own search time, measured on the
            # device (end of the target selection to the start of the update; the reference brackets
            # _find_source_patch with time.time(), :64-69).
            # show_progress (:55-57): frame c is the working image before iteration c's update, i.e.
            # the input with the first c patch copies of the trace applied -- rebuilt on the host
            # and encoded by worker threads while the GPU runs the next chunk.
            frames = _FrameDumper(img, mask_u8, self.patch_size) if self.show_progress else None
            shown = 0
            try:
                while True:
                    _, remaining = eng.run(max_iters=self.VERBOSE_CHUNK)
                    recs = eng.trace(0)
                    found_us = eng.search_times(0) if self.verbose else None
                    self.search_seconds = None if found_us is None else found_us.astype(np.float64) * 1e-6
                    for k in range(shown, len(recs)):
                        r = recs[k]
                        if frames is not None:
                            frames.emit_then_apply(r)
                        if self.verbose:
                            print('Time to find best: %.2f seconds' % (1e-6 * float(found_us[k])))
                            print('%d of %d completed' % (total - int(r['remaining']), total))
                    shown = len(recs)
                    if remaining == 0:
                        break
            finally:
                if frames is not None:
                    frames.close()
        else:
            eng.run()

        out, conf, msk = eng.download(want_conf=True, want_mask=True)
        self.working_image = out[0]
        self.confidence = conf[0]
        self.working_mask = msk[0].astype(self.working_mask.dtype)
        self.iterations = eng.iterations(0)
        self._fields_stale = True

        if self.verbose:
            print('Inpainting took %.2f seconds to complete' % (time.time() - start_time))
        return self.working_image

    def trace(self):
        """Per-iteration (target, source, cost, remaining) records of the last inpaint()."""
        if self._engine is None:
            raise RuntimeError('inpaint() has not been run')
        return self._engine.trace(0)

    def _validate_inputs(self):
        if self.image.shape[:2] != self.mask.shape:
            raise AttributeError('mask and image must be of the same size')

    def __del__(self):
        eng = getattr(self, '_engine', None)
        if eng is not None:
            eng.close()


# north-star spelling
CriminisiInpainter = Criminisi_Inpainter


def inpaint_image(image, patch_size=9, dilation_size=4, device=0):
    """Mirror of /root/reference/src/utils.py:67-103 (`inpaint_image`): builds the mask of missing
    (near-black) pixels and inpaints it, with the mask built on the GPU as well."""
    img = np.asarray(image)
    if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
        raise ValueError('image must be a uint8 array of shape (H, W, 3) in BGR order')
    with capi.Engine(img.shape[0], img.shape[1], patch_size, batch=1, device=device) as eng:
        eng.upload_automask(img, dilation_size=dilation_size)
        eng.run()
        out, _, _ = eng.download()
    return out[0]


def filter_noise(image, median_ksize=3, nlm_h=7, apply_sharpening=True, device=0):
    """Mirror of /root/reference/src/utils.py:106-138 (`filter_noise`): median blur, non-local
    means in Lab (h = nlm_h on L, 1 on a and b, template 9, search 31) and Laplacian sharpening,
    all on the GPU."""
    return capi.filter_noise(image, median_ksize=median_ksize, nlm_h=nlm_h,
                             apply_sharpening=apply_sharpening, device=device)


def inpaint_many(images, masks, patch_size=9, device=0, batch=None):
    """Inpaints a list/array of same-shaped images as lock-step batches on one GPU: the loop of
    /root/reference/scripts/main.py:44-58 around Criminisi_Inpainter(...).inpaint(), without the
    per-image launch latency.  Returns uint8 [n,H,W,3]."""
    images = np.ascontiguousarray(images, dtype=np.uint8)
    masks = np.ascontiguousarray(np.asarray(masks) != 0, dtype=np.uint8)
    n, H, W = masks.shape
    if images.shape != (n, H, W, 3):
        raise AttributeError('mask and image must be of the same size')
    batch = n if batch is None else max(1, min(int(batch), n))
    out = np.empty_like(images)
    with capi.Engine(H, W, patch_size, batch=batch, device=device) as eng:
        for lo in range(0, n, batch):
            hi = min(lo + batch, n)
            cnt = hi - lo
            if cnt < batch:
                # pad the last batch with empty masks (finish after one no-op iteration)
                pad_i = np.zeros((batch - cnt, H, W, 3), np.uint8)
                pad_m = np.zeros((batch - cnt, H, W), np.uint8)
                eng.upload(np.concatenate([images[lo:hi], pad_i]), np.concatenate([masks[lo:hi], pad_m]))
            else:
                eng.upload(images[lo:hi], masks[lo:hi])
            eng.run()
            res, _, _ = eng.download(0, cnt)
            out[lo:hi] = res
    return out
