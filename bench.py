#!/usr/bin/env python3
"""Benchmark of the Criminisi hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--configs 0,1,2,3,4]

Headline workload (BASELINE.json configs[1]): 256x256 grey X-ray-like images replicated to 3
channels, a synthetic 10 % mask, patch_size 9.  One GPU fills a batch of B such images in lock
step; a "step" is one complete pass of the fill loop over that batch (every image inpainted from
its original mask to an empty mask), starting from inputs already resident in HBM.  With N > 1
every rank runs its own batch on its own GPU (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0: value = inpainted images / s over all GPUs; `value_incl_host_io`
= the same batch streamed from and to page-locked host memory (two half-batch engines on two
host threads, so that one half's PCIe transfers run under the other half's kernels);
`evals_per_s_per_gpu` is the second half of the metric (masked patch-SSD window evaluations per
second of search-kernel time on one GPU); `roofline` prices the dominant kernel (the search);
`cpu_baseline` times the C oracle ("port") on the host cores over a bounded sample of the same
images; `configs` holds one entry per BASELINE.json config (the 100 dataset images, the headline,
800 x 512^2, 2048^2 with patch 9 and 15 run to completion, 4096^2), each with its own throughput,
iteration counts, evals/s and roofline of its search kernel.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "image-processing_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

H = W = 256
PATCH = 9
MASK_FRACTION = 0.10
CONFIG_ID = 2                      # seed = 1000 * config + image index
MFMA_I8_PEAK_TOPS = 5033.2         # 256 CU x 4 SIMD x 1024 MAC/clk x 2 op x 2.4 GHz (dense i8 MFMA)
HBM_PEAK_GBS = 8000.0              # HBM3E


def _make_one(args):
    import crim_synth
    h, w, seed, colour = args
    return (crim_synth.xray_like(h, w, seed, colour=colour),
            crim_synth.blob_mask(h, w, seed, fraction=MASK_FRACTION, margin=PATCH))


def host_cores():
    """Cores this process may actually use: the affinity mask, further limited by OMP_NUM_THREADS
    when the box sets it (a shared host shows every CPU of the machine in os.cpu_count())."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 1:                      # (1 is what torch.distributed.run sets by default: not a core count)
        n = min(n, env)
    elif env == 1 and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        n = max(1, min(n, 16) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))) or 1)
        n = max(n, 2)
    return max(1, n)


def make_batch(batch, h, w, config_id, first_index, colour, indices=None):
    """Seeded synthetic images and masks (seed = 1000 * config + global image index), generated on
    a few host processes (forked before this process touches the GPU)."""
    import multiprocessing as mp
    if indices is None:
        indices = range(first_index, first_index + batch)
    tasks = [(h, w, 1000 * config_id + i, colour) for i in indices]
    batch = len(tasks)
    imgs = np.empty((batch, h, w, 3), np.uint8)
    masks = np.empty((batch, h, w), np.uint8)
    workers = max(1, min(host_cores(), batch // 8))
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_make_one, tasks, chunksize=max(1, min(16, batch // (4 * workers))))
    else:
        res = [_make_one(t) for t in tasks]
    for i, (im, mk) in enumerate(res):
        imgs[i] = im
        masks[i] = mk
    return imgs, masks


def pmc_traffic_per_launch(images_per_launch):
    """HBM-side bytes per search launch from the committed rocprofv3 PMC passes of the default
    command (FETCH_SIZE and WRITE_SIZE are collected in separate runs, tools/pmc_passes.sh; they
    cannot be read from inside the benchmark), scaled to this run's images per launch.
    FETCH_SIZE is doubled: the search stages its tiles with 16-byte-per-lane coalesced loads
    (straight into LDS), which gfx950 tallies at half their bytes.  None when no PMC summary is
    present."""
    path = os.path.join(ROOT, "profiles", "pmc_search_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        per_launch = (2.0 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024.0
        return per_launch * images_per_launch / float(d.get("images_per_launch", 2048))
    except (OSError, KeyError, ValueError):
        return None


def stencil_roofline():
    """The stencil stages (front / confidence / data / priority: k_priority; update: k_update) of the
    headline batch against the HBM roof: (FETCH_SIZE + WRITE_SIZE) per launch from the committed rocprofv3
    PMC passes over the average launch duration of the kernel trace of the same command
    (profiles/pmc_stencil_latest.json, written by tools/pmc_passes.sh; counters cannot be read from inside
    the benchmark).  None when no summary is present."""
    path = os.path.join(ROOT, "profiles", "pmc_stencil_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    out = {"source": "profiles/pmc_stencil_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, "
                     "--kernel-trace --stats for the durations; FETCH_SIZE as counted, in KB of 1024 B)",
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "bound": "hbm"}
    for k, v in d.get("kernels", {}).items():
        try:
            byts = (v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024.0
            gbs = byts / (v["avg_launch_us"] * 1e-6) / 1e9
            out[k] = {"traffic": byts, "avg_launch_us": v["avg_launch_us"], "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                      "share_of_gpu_time": v.get("share_of_gpu_time"), "images_per_launch": v.get("images_per_launch"),
                      "salu_per_valu": v.get("salu_per_valu")}
        except (KeyError, ZeroDivisionError):
            continue
    return out


def launch_split(capi, h, w, patch, upload, device, iters):
    """Where an iteration's microseconds go, from three device time stamps per iteration (start of the launch
    that selects and searches, end of the selection, start of the launch that follows the search): a
    traced run of `iters` iterations on an engine of its own."""
    with capi.Engine(h, w, patch, device=device) as e:
        upload(e)
        e.run(max_iters=64)
        e.run(max_iters=iters)
        st = e.iteration_stamps(0)[64:].astype(np.float64) * 0.01          # microseconds
    if len(st) < 8:
        return None
    search = st[:, 2] - st[:, 0]                       # selecting launch start -> following launch start
    rest = st[1:, 0] - st[:-1, 2]                      # following launch start -> next selecting launch start
    ok = rest < 10 * np.median(rest)                   # (graph boundaries and host rounds excluded)
    return {"search_launch_us": float(np.median(search)), "of_which_selection_us": float(np.median(st[:, 1] - st[:, 0])),
            "update_and_next_priority_launch_us": float(np.median(rest[ok])),
            "iteration_us_by_stamps": float(np.median((st[1:, 0] - st[:-1, 0])[ok])),
            "note": "medians over %d iterations; each figure runs from the start of one launch to the start of the "
                    "next, so launch boundaries are included; an iteration is two launches: the search (which "
                    "also selects the target) and k_step (update + the next iteration's priority stage)" % len(st)}


def _cpu_one(args):
    from oracle import c_oracle
    img, mask = args
    t = time.time()
    out, conf, msk, tr, n = c_oracle.run(img, mask, PATCH, want_trace=False)
    import zlib
    return time.time() - t, n, zlib.crc32(out.tobytes())


def cpu_baseline(imgs, masks, budget_images_per_core=4):
    """C oracle over a bounded sample, one image per task on a process pool."""
    import multiprocessing as mp
    from oracle import c_oracle
    c_oracle._lib()
    cores = host_cores()
    n = min(len(imgs), cores * budget_images_per_core)
    tasks = [(imgs[i], masks[i]) for i in range(n)]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_one, tasks)
    wall = time.time() - t0
    iters = sum(r[1] for r in res)
    return {"_crc": [r[2] for r in res],
            "value": n / wall, "unit": "images/s", "cores": cores, "kind": "port",
            "host_cpus_visible": os.cpu_count(),
            "sample": "%d of the batch's 256x256 images (first by index), C oracle "
                      "oracle/criminisi_oracle.c, one image per process, %d processes = every core this "
                      "job may use (affinity mask and OMP_NUM_THREADS; os.cpu_count() shows %s for the "
                      "whole machine); %.1f s wall, %d iterations" % (n, cores, os.cpu_count(), wall, iters)}


def _cpu_crc(args):
    from oracle import c_oracle
    import zlib
    img, mask = args
    return zlib.crc32(c_oracle.run(img, mask, PATCH, want_trace=False)[0].tobytes())


def cpu_outputs_crc(imgs, masks, count):
    """CRC-32 of the C oracle's output for the first `count` images (one process each)."""
    import multiprocessing as mp
    from oracle import c_oracle
    c_oracle._lib()
    with mp.get_context("fork").Pool(min(host_cores(), count)) as pool:
        return pool.map(_cpu_crc, [(imgs[i], masks[i]) for i in range(count)])


def single_image(capi, device, img, mask):
    """configs[1] as BASELINE words it: ONE 256x256 image, start to finish (upload, fill, download)."""
    with capi.Engine(H, W, PATCH, device=device, flags=capi.FLAG_NO_TRACE) as eng:
        eng.upload(img, mask)
        eng.run()                              # builds the graphs
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            eng.upload(img, mask)
            it, rem = eng.run()
            out = eng.download()[0][0]
            ts.append(time.perf_counter() - t0)
        t_fill = []
        for _ in range(5):
            eng.upload(img, mask)
            t0 = time.perf_counter()
            it, rem = eng.run()
            t_fill.append(time.perf_counter() - t0)
    dt, df = float(np.median(ts)), float(np.median(t_fill))
    return {"what": "one 256x256 grey image of the batch alone on the GPU (batch of 1), host arrays in and out; "
                    "median of 5", "seconds": dt, "images_per_s": 1.0 / dt, "iterations": int(it),
            "fill_loop_seconds": df, "us_per_iteration": 1e6 * df / max(int(it), 1),
            "launches_us": launch_split(capi, H, W, PATCH, lambda e: e.upload(img, mask), device, 200)}, out


def directory_driver_runs():
    """JPEG in -> JPEG out through the directory driver (the loop of scripts/main.py:41-60), in processes of
    their own: the driver forks its JPEG workers before it touches the GPU."""
    import subprocess
    out = {}
    for key, extra in (("dataset_100_images", ["--dataset"]), ("synthetic_2048_images", ["--images", "2048", "--batch", "512"])):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_directory.py")] + extra,
                               capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[key] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-400:]}
        except Exception as e:                              # noqa: BLE001 (a side measurement must not end the run)
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    out["note"] = ("crim_batch_driver.inpaint_directory: list *.jpg, decode (Pillow, worker processes), mask on the "
                   "GPU, fill in lock-step batches, encode and write under the same names; seconds cover the whole "
                   "call incl. engine creation and graph capture; dataset_100_images is the reference's own set")
    return out


def cpu_seconds_per_iteration(img, mask, patch, iters=2):
    """The C oracle's first `iters` iterations on one large image (a full CPU run takes hours)."""
    from oracle import c_oracle
    t = time.time()
    n = c_oracle.run(img, mask, patch, max_iters=iters, want_trace=False)[4]
    return (time.time() - t) / max(n, 1)


# --------------------------------------------------------------------------------------------
# roofline helpers
# --------------------------------------------------------------------------------------------

def search_roofline(capi, h, w, patch, batch, upload, kernel, planes, hbm_resident, device, max_iters=0,
                    traffic=None):
    """One more pass (bounded by max_iters) on an engine that brackets every search launch with HIP
    events on the engine's own stream, each bracketed launch alone on the GPU.  `planes` byte planes
    plus the 1/8-byte validity bitmap are the algorithmic bytes per pixel, image and iteration."""
    with capi.Engine(h, w, patch, batch=batch, device=device,
                     flags=capi.FLAG_PROFILE_SOLO | capi.FLAG_NO_TRACE) as pe:
        upload(pe)
        pe.run(max_iters)
        ms, launches, evals = pe.profile()
    if ms <= 0.0 or launches <= 0:
        # (a side pass that recorded nothing must not cost the whole line; the figure is then missing, visibly)
        return {"kernel": kernel, "error": "the profiling pass recorded no search launch (%r ms, %r launches)" % (ms, launches),
                "bound": "hbm" if hbm_resident else "mfma", "achieved": None, "peak": None, "unit": None, "frac": None,
                "avg_launch_us": None, "traffic": traffic}, 0.0
    P = 2 * ((patch - 1) // 2) + 1
    windows = (h - P + 1) * (w - P + 1)
    bytes_per_window = (planes + 0.125) * h * w / windows
    algo_bytes = evals * bytes_per_window
    gbs = algo_bytes / (ms * 1e-3) / 1e9
    tops = evals * P * P * 3 * 2 / (ms * 1e-3) / 1e12
    r = {"kernel": kernel, "avg_launch_us": 1e3 * ms / max(launches, 1), "launches": launches,
         "algorithmic_bytes_per_launch": algo_bytes / max(launches, 1),
         "hbm_equiv_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
         "mfma_achieved_tops": tops, "mfma_peak_tops": MFMA_I8_PEAK_TOPS, "mfma_frac": tops / MFMA_I8_PEAK_TOPS,
         "traffic": traffic}
    if hbm_resident:
        r.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
    else:
        # the planes of these images stay in the 256 MB Infinity Cache between iterations: the matrix
        # cores are the roof that applies (243 or 675 algorithmic MAC per window)
        r.update({"bound": "mfma", "achieved": tops, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                  "frac": tops / MFMA_I8_PEAK_TOPS})
    return r, evals / (ms * 1e-3)


# --------------------------------------------------------------------------------------------
# per-config measurements (rank 0, single-GPU runs)
# --------------------------------------------------------------------------------------------

def config_dataset(capi, device, steps):
    """configs[0]: the reference's 100 dataset images (scripts/main.py loop), patch 9, one lock-step
    batch; host arrays in, host arrays out."""
    import zlib
    z = np.load(os.path.join(ROOT, "tests", "golden", "xray_images", "decoded_bgr.npz"))
    names = [str(n) for n in z["names"]]
    imgs = z["bgr"]                    # the JPEGs as Pillow decoded them when the expected results were made
    with open(os.path.join(ROOT, "tests", "golden", "xray_dataset_expected.json")) as f:
        want = json.load(f)
    B = len(names)
    pin_in = capi.pinned_empty(imgs.shape)
    pin_in[:] = imgs
    imgs = pin_in
    out = capi.pinned_empty(imgs.shape)
    try:
        with capi.Engine(256, 256, 9, batch=B, device=device, flags=capi.FLAG_NO_TRACE) as eng:
            def step():
                eng.upload_automask(imgs, dilation_size=4)
                eng.run()
                eng.download(out=out)
            step()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            dt = (time.perf_counter() - t0) / steps
            iters = np.array([eng.iterations(i) for i in range(B)])
        same = sum(int(zlib.crc32(out[i].tobytes()) == want[n]["crc32_out"] and int(iters[i]) == want[n]["iterations"])
                   for i, n in enumerate(names))
        roof, evals_s = search_roofline(capi, 256, 256, 9, B, lambda e: e.upload_automask(imgs, dilation_size=4),
                                        "k_search_pipe<9, true>", 5, False, device)
        # the stage that follows the inpainter in the reference's pipeline (filter_noise, utils.py:106-138)
        capi.filter_noise(out[:4])
        t0 = time.perf_counter()
        capi.filter_noise(out, median_ksize=3, nlm_h=7, apply_sharpening=True)
        nf_dt = time.perf_counter() - t0
    finally:
        capi.pinned_free(imgs)
        capi.pinned_free(out)
    return {"workload": "BASELINE configs[0]: the reference's %d dataset images (256x256, false colour, decoded "
                        "with Pillow), mask built on the GPU by the recipe of src/utils.py:79-100 without the "
                        "perspective stage (it needs OpenCV), patch_size 9, one lock-step batch; time covers host "
                        "-> device, mask construction, the fill loop and device -> host (no JPEG coding)" % B,
            "images": B, "images_per_s_incl_host_io": B / dt, "ms_per_batch": 1e3 * dt,
            "iterations_per_image_mean": float(iters.mean()), "iterations_per_image_max": int(iters.max()),
            "us_per_lockstep_iteration": 1e6 * dt / int(iters.max()),
            "outputs_identical_to_cpu_oracle": "%d of %d (CRC-32 of the output image and iteration count against "
                                               "tests/golden/xray_dataset_expected.json, checked outside the timed region)" % (same, B),
            "evals_per_s_per_gpu": evals_s, "roofline": roof,
            "next_stage_filter_noise": {
                "what": "filter_noise(median 3, non-local means h=7 / hColor=1, template 9, search 31, sharpening) "
                        "of src/utils.py:106-138 on the %d inpainted images, host arrays in and out" % B,
                "images_per_s_incl_host_io": B / nf_dt, "ms_per_batch": 1e3 * nf_dt,
                # per pixel 2 runs x 961 candidates x 81 squared differences (two 16-bit lanes per dword)
                "window_differences_per_s": B * 256 * 256 * 961 * 81 * 3 / nf_dt}}


def config_512_batch(capi, torch, device, imgs, masks, steps, oracle_crcs=None):
    """configs[2] on one GPU: 800 synthetic 512x512 false-colour images, 10 % masks, patch 9."""
    B = len(imgs)
    d_img = torch.from_numpy(imgs).cuda()
    d_mask = torch.from_numpy(masks).cuda()
    d_out = torch.empty_like(d_img)
    torch.cuda.synchronize()
    with capi.Engine(512, 512, 9, batch=B, device=device, flags=capi.FLAG_NO_TRACE) as eng:
        def step():
            eng.upload_device(d_img.data_ptr(), d_mask.data_ptr())
            eng.run()
            eng.download_device(d_out.data_ptr())
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        iters = np.array([eng.iterations(i) for i in range(B)])
        check = None
        if oracle_crcs:
            import zlib
            got = eng.download(0, len(oracle_crcs))[0]
            check = "%d of %d" % (sum(int(zlib.crc32(got[i].tobytes()) == c) for i, c in enumerate(oracle_crcs)),
                                  len(oracle_crcs))
    roof, evals_s = search_roofline(
        capi, 512, 512, 9, B, lambda e: e.upload_device(d_img.data_ptr(), d_mask.data_ptr()),
        "k_search_pipe<9, true>", 5, True, device, max_iters=200)
    del d_img, d_mask, d_out
    return {"workload": "BASELINE configs[2] on one GPU: %d synthetic 512x512 false-colour images, 10 %% masks, "
                        "patch_size 9, one lock-step batch, inputs and outputs resident in HBM" % B,
            "images": B, "images_per_s": B / dt, "ms_per_batch": 1e3 * dt,
            "iterations_per_image_mean": float(iters.mean()), "iterations_per_image_max": int(iters.max()),
            "outputs_identical_to_cpu_oracle": check,
            "outputs_check_note": "the first images of the batch, filled by the C oracle on the host cores before the "
                                  "GPU was touched; compared outside the timed region",
            "evals_per_s_per_gpu": evals_s, "roofline": roof,
            "roofline_note": "search launches of the first 200 lock-step iterations, each alone on the GPU"}


def sharded_512(capi, torch, dist, device, imgs, masks, world, backend):
    """configs[2] as BASELINE words it: a FIXED set of 800 512x512 images divided among the ranks
    (image i on rank i mod N, no data-path collective); every rank fills its share, the slowest
    rank's time counts (strong scaling)."""
    n = len(imgs)
    d_img = torch.from_numpy(imgs).cuda()
    d_mask = torch.from_numpy(masks).cuda()
    d_out = torch.empty_like(d_img)
    with capi.Engine(512, 512, 9, batch=n, device=device, flags=capi.FLAG_NO_TRACE) as eng:
        def step():
            eng.upload_device(d_img.data_ptr(), d_mask.data_ptr())
            eng.run()
            eng.download_device(d_out.data_ptr())
        step()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    return {"workload": "BASELINE configs[2]: a fixed set of 800 synthetic 512x512 false-colour images, 10 %% masks, "
                        "patch_size 9, image i on rank i mod %d; inputs and outputs resident in HBM" % world,
            "images": 800, "ranks": world, "images_on_this_rank": n, "scaling": "strong",
            "images_per_s": 800.0 / dt, "ms": 1e3 * dt}


def config_large(capi, device, side, patch, config_index, profile_iters=1500):
    """configs[3] / configs[4]: one side x side false-colour image with a centred hole of a quarter
    of the frame, run to completion."""
    import crim_synth
    img = crim_synth.xray_like(side, side, 4000 + side, colour=True)
    mask = crim_synth.square_hole_mask(side, side, side // 2)
    cpu_s = cpu_seconds_per_iteration(img, mask, patch)
    with capi.Engine(side, side, patch, device=device, flags=capi.FLAG_NO_TRACE) as eng:
        eng.upload(img, mask)
        eng.run(max_iters=64)              # builds the graphs, warms the caches
        eng.upload(img, mask)
        t0 = time.perf_counter()
        it, rem = eng.run()
        dt = time.perf_counter() - t0
        out = eng.download()[0][0]
    ok = bool(rem == 0 and np.array_equal(out[mask == 0], img[mask == 0]))
    P = 2 * ((patch - 1) // 2) + 1
    windows = (side - P + 1) ** 2
    roof, evals_s = search_roofline(capi, side, side, patch, 1, lambda e: e.upload(img, mask),
                                    "k_search_pipe<%d, true>" % P, 5, False, device, max_iters=profile_iters)
    split = launch_split(capi, side, side, patch, lambda e: e.upload(img, mask), device, 3000)
    return {"workload": "BASELINE configs[%d]: one %dx%d false-colour image, centred %dx%d hole (25 %%), "
                        "patch_size %d, filled to completion" % (config_index, side, side, side // 2, side // 2, patch),
            "iterations": int(it), "seconds": dt, "images_per_s": 1.0 / dt,
            "us_per_iteration": 1e6 * dt / max(it, 1),
            "launches_us": split,
            "evals_per_s_whole_iteration": windows * it / dt,
            "evals_per_s_per_gpu": evals_s,
            "search_share_of_iteration": (roof["avg_launch_us"] / (1e6 * dt / max(it, 1))) if roof.get("avg_launch_us") else None,
            "completed_and_known_pixels_untouched": ok,
            "cpu_port_s_per_iteration": cpu_s,
            "cpu_port_note": "C oracle, first 2 iterations, one core; a full CPU run would take "
                             "%.1f hours at that rate" % (cpu_s * it / 3600.0),
            "roofline": roof,
            "roofline_note": "search launches of the first %d iterations, each alone on the GPU" % profile_iters}


# --------------------------------------------------------------------------------------------
# host-I/O-inclusive throughput of the headline batch
# --------------------------------------------------------------------------------------------

def streamed_images_per_s(capi, device, imgs, masks, steps, shares=4, engines_n=5):
    """The headline batch from and to page-locked host memory.  The batch is cut into `shares`
    shares; `engines_n` engines of one share each (one host thread per engine) take shares from a
    queue and loop upload -> fill -> download.  One transfer phase at a time (a lock), and one engine
    more than there are shares in a batch: while an engine's transfers run, a batch's worth of images
    is still being filled by the others."""
    import queue
    B = len(imgs)
    part = B // shares
    pin_img, pin_msk, out = (capi.pinned_empty(imgs.shape), capi.pinned_empty(masks.shape),
                             capi.pinned_empty(imgs.shape))
    pin_img[:] = imgs
    pin_msk[:] = masks
    imgs, masks = pin_img, pin_msk
    engines = [capi.Engine(H, W, PATCH, batch=part, device=device, flags=capi.FLAG_NO_TRACE | capi.FLAG_ONE_LANE)
               for _ in range(engines_n)]
    errors = []
    link = threading.Lock()

    def loop(k, work):
        try:
            while True:
                try:
                    share = work.get_nowait()
                except queue.Empty:
                    return
                lo = share * part
                with link:
                    engines[k].upload(imgs[lo:lo + part], masks[lo:lo + part])
                engines[k].run()
                with link:
                    engines[k].download(out=out[lo:lo + part])
        except Exception as e:            # noqa: BLE001 (reported to the main thread)
            errors.append(e)

    def timed(passes):
        work = queue.Queue()
        for _ in range(passes):
            for sh in range(shares):
                work.put(sh)
        ts = [threading.Thread(target=loop, args=(k, work)) for k in range(engines_n)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    try:
        timed(2)
        dt = timed(steps)
        if errors:
            raise errors[0]
    finally:
        for e in engines:
            e.close()
    result = out[:64].copy()
    for a in (imgs, masks, out):
        capi.pinned_free(a)
    return shares * part * steps / dt, result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--configs", default="0,1,2,3,4",
                    help="BASELINE.json configs to measure besides the headline (single-GPU runs; '' for none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-io", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="skip the RCCL split-search workers of configs[4]")
    ap.add_argument("--split-iters", type=int, default=2000)
    ap.add_argument("--colour", action="store_true",
                    help="false-colour images (5-plane search) instead of the grey ones of BASELINE configs[1]; "
                         "side measurement, run with --no-roofline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    want = set(int(c) for c in args.configs.split(",") if c.strip() != "") if world == 1 else set()
    B = args.batch

    # The intra-image split search of configs[4] drives RCCL through the engine library, which must
    # not share a process with PyTorch's bundled ROCm runtime: one worker process per rank, started
    # here -- before this process touches the GPU -- and released through a file further down.
    split_proc = None
    split_dir = None
    # (under a profiler the process must stay alone: its children would inherit the profiler's preload)
    profiled = any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if (world > 1 or 4 in want) and not args.no_split and not profiled:
        import crim_split_worker as sw
        split_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"),
                                 "crim_split_%s_%s" % (os.environ.get("MASTER_PORT", "solo"),
                                                       os.environ.get("TORCHELASTIC_RUN_ID", str(os.getppid()))))
        if rank == 0:
            import shutil
            shutil.rmtree(split_dir, ignore_errors=True)
        os.makedirs(split_dir, exist_ok=True)
        dev = int(os.environ.get("BENCH_DEVICE", local_rank))
        # (the one-GPU rehearsal of a multi-rank launch -- BENCH_BACKEND=gloo -- exchanges through host memory:
        # RCCL refuses two ranks on one device)
        split_proc = sw.launch_worker(rank, world, dev, split_dir, size=4096, patch=9, iters=args.split_iters, wait=600.0,
                                      transport="host" if os.environ.get("BENCH_BACKEND") == "gloo" else "rccl")
    try:
        run_benchmark(args, rank, local_rank, world, distributed, want, B, split_proc, split_dir, profiled)
    finally:
        # a worker that was never released (an error above) must not sit on the GPU until its own time-out
        if split_proc is not None and split_proc.poll() is None:
            try:
                split_proc.wait(timeout=5 if sys.exc_info()[0] is not None else 270)
            except Exception:                                   # noqa: BLE001
                split_proc.kill()


def run_benchmark(args, rank, local_rank, world, distributed, want, B, split_proc, split_dir, profiled):
    imgs, masks = make_batch(B, H, W, CONFIG_ID, rank * B, args.colour)
    imgs512 = masks512 = None
    if 2 in want:
        imgs512, masks512 = make_batch(800, 512, 512, 3, 0, True)
    elif world > 1:
        imgs512, masks512 = make_batch(0, 512, 512, 3, 0, True, indices=range(rank, 800, world))

    # CPU baseline first (rank 0, single-GPU runs only), before this process touches the GPU:
    # the worker pool forks.
    cpu = None
    crc512 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(imgs, masks)
        if imgs512 is not None:
            crc512 = cpu_outputs_crc(imgs512, masks512, 4)

    import torch
    import torch.distributed as dist
    import crim_hip_capi as capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    # Rehearsal knobs (several ranks on a box with fewer GPUs): BENCH_DEVICE pins every rank to one
    # device, BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one GPU.
    if "BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_DEVICE"])
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # inputs resident in HBM before the timed region
    d_img = torch.from_numpy(imgs).cuda()
    d_mask = torch.from_numpy(masks).cuda()
    d_out = torch.empty_like(d_img)
    torch.cuda.synchronize()

    eng = capi.Engine(H, W, PATCH, batch=B, device=local_rank, flags=capi.FLAG_NO_TRACE)

    def step():
        eng.upload_device(d_img.data_ptr(), d_mask.data_ptr())
        eng.run()
        eng.download_device(d_out.data_ptr())

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    iters_per_image = np.array([eng.iterations(i) for i in range(B)])
    total_images = B * world * args.steps
    value = total_images / elapsed

    result = {
        "metric": "inpainted images/sec (node) + masked patch-SSD evals/sec per GPU",
        "value": value,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i8",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 256x256 grey X-ray-like image, synthetic "
                               "10%% mask, patch_size 9; %d such images per GPU filled in lock "
                               "step (one step = all of them start to finish)" % B,
                   "batch_per_gpu": B, "height": H, "width": W, "patch_size": PATCH,
                   "mask_fraction": MASK_FRACTION,
                   "iterations_per_image_mean": float(iters_per_image.mean()),
                   "iterations_per_image_max": int(iters_per_image.max())},
    }
    gpu_first = eng.download(0, min(B, 64))[0] if cpu is not None else None
    eng.close()

    if rank == 0 and not args.no_roofline:
        # Same batch once more on an engine that brackets every search launch with HIP events
        # on the engine's own streams (eager launches; the timed region above replays graphs),
        # each bracketed launch alone on the GPU: the kernel's own duration.  In the timed region
        # the two halves of the batch run on two streams and mostly take turns on the GPU, so a
        # launch there also waits for, or shares the CUs with, the other half's kernels; the
        # second pass below brackets the launches under those conditions (stream-ordered events:
        # the wait is included).
        def up(e):
            e.upload_device(d_img.data_ptr(), d_mask.data_ptr())
        roof, evals_s = search_roofline(capi, H, W, PATCH, B, up,
                                        "k_search_pipe<9, %s>" % ("true" if args.colour else "false"),
                                        5 if args.colour else 3, True, local_rank,
                                        traffic=pmc_traffic_per_launch(B / 2.0 if B >= 32 else B))
        with capi.Engine(H, W, PATCH, batch=B, device=local_rank, flags=capi.FLAG_PROFILE | capi.FLAG_NO_TRACE) as pe:
            up(pe)
            pe.run()
            ov_ms, ov_launches, _ = pe.profile()
        roof["overlapped_avg_launch_us"] = 1e3 * ov_ms / max(ov_launches, 1)
        roof["note"] = ("achieved/frac/avg_launch_us: every search launch of one pass over the batch, each "
                        "alone on the GPU; overlapped_avg_launch_us: the same with the two halves of the "
                        "batch on two streams as in the timed region (event to event, waiting for the other "
                        "half's kernels included); algorithmic bytes = 3.125 B per pixel per image per "
                        "iteration (3 byte planes + validity bitmap), one launch covers one lane = half the "
                        "batch; traffic = HBM-side bytes per launch from the committed PMC passes "
                        "(profiles/; FETCH_SIZE doubled for 16-byte loads, plus WRITE_SIZE); mfma_*: integer "
                        "ops (2 per MAC) against the dense i8 MFMA peak, 243 algorithmic MAC per window")
        result["evals_per_s_per_gpu"] = evals_s
        result["roofline"] = roof
        result["roofline_stencil"] = stencil_roofline()
    del d_img, d_mask, d_out
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_host_io:
        v, streamed_out = streamed_images_per_s(capi, local_rank, imgs, masks, args.steps)
        result["value_incl_host_io"] = v
        result["host_io_note"] = ("same batch from and to page-locked host memory (H2D 262144 B + D2H 196608 B per "
                                  "image over PCIe): the batch in four shares of %d images, five engines of one share each on five "
                                  "host threads taking shares from a queue (upload -> fill -> download, one transfer "
                                  "phase at a time), %d passes over the batch" % (B // 4, args.steps))
        if gpu_first is not None:
            result["host_io_outputs_identical"] = bool(np.array_equal(streamed_out[:len(gpu_first)], gpu_first))
        del streamed_out

    if cpu is not None:
        # the sample's outputs double as a check of the timed run's results (outside the timed region)
        import zlib
        crcs = cpu.pop("_crc")
        same = sum(int(zlib.crc32(gpu_first[i].tobytes()) == c) for i, c in enumerate(crcs) if i < len(gpu_first))
        cpu["outputs_identical_to_gpu"] = "%d of %d" % (same, min(len(crcs), len(gpu_first)))
        result["cpu_baseline"] = cpu

    if rank == 0 and world == 1 and want:
        first_img, first_mask = imgs[0].copy(), masks[0].copy()
        del imgs, masks
        cfgs = {}
        if 0 in want:
            cfgs["configs[0]"] = config_dataset(capi, local_rank, args.steps)
        if 1 in want:
            one, one_out = single_image(capi, local_rank, first_img, first_mask)
            one["output_identical_to_batch_run"] = bool(gpu_first is not None and np.array_equal(one_out, gpu_first[0])) \
                if gpu_first is not None else None
            cfgs["configs[1]"] = {"workload": result["config"]["workload"], "images_per_s": value,
                                  "single_image": one,
                                  "images_per_s_incl_host_io": result.get("value_incl_host_io"),
                                  "iterations_per_image_mean": float(iters_per_image.mean()),
                                  "evals_per_s_per_gpu": result.get("evals_per_s_per_gpu"),
                                  "roofline": {k: result["roofline"][k] for k in
                                               ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us")}
                                  if "roofline" in result else None}
        if 2 in want:
            cfgs["configs[2]"] = config_512_batch(capi, torch, local_rank, imgs512, masks512, max(1, args.steps - 1), crc512)
            del imgs512, masks512
            torch.cuda.empty_cache()
        if 3 in want:
            cfgs["configs[3] patch 9"] = config_large(capi, local_rank, 2048, 9, 3)
            cfgs["configs[3] patch 15"] = config_large(capi, local_rank, 2048, 15, 3)
        if 4 in want:
            cfgs["configs[4]"] = config_large(capi, local_rank, 4096, 9, 4)
            cfgs["configs[4]"]["multi_gpu_note"] = ("one GPU; the intra-image split over N GPUs (one in-place RCCL all-gather of the "
                                                    "ranks' search-slot shares per iteration, 32 KB in all at this size) is "
                                                    "measured by `bench.py --gpus N`")
        if 0 in want and not profiled:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            cfgs["configs[0]"]["directory_driver"] = directory_driver_runs()
        result["configs"] = cfgs

    if distributed:
        # every rank takes part: the fixed 800-image set of configs[2], sharded
        sh = sharded_512(capi, torch, dist, local_rank, imgs512, masks512, world, backend)
        if rank == 0:
            result.setdefault("configs", {})["configs[2] sharded"] = sh

    if split_proc is not None:
        import crim_split_worker as sw
        # this process's GPU work is over: let the split workers have the GPUs
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if distributed:
            dist.barrier()
        if rank == 0:
            sw.release(split_dir)
            res = sw.summarise(sw.collect(split_dir, world, 240.0))
            res["note"] = ("configs[4], %d rank(s): every rank holds a replica of the 4096x4096 image, scores its share of "
                           "the window-origin rows and the ranks' shares of the per-block best windows (16 B per search "
                           "block) are put together by one in-place RCCL all-gather per iteration, captured with the "
                           "iteration's kernels into the replayed graphs (graph_captured); us_per_iteration is the slowest rank's, over %d iterations after "
                           "32 warm-up ones; single_gpu_us_per_iteration is the unsplit engine on rank 0's GPU over the "
                           "same iterations (the crossover is where the former drops below the latter)"
                           % (world, args.split_iters))
            result.setdefault("configs", {}).setdefault("configs[4]", {})["split_rccl"] = res
        if distributed:
            dist.barrier()

    if rank == 0:
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
