#!/usr/bin/env python3
"""Benchmark of the Criminisi hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

Workload (BASELINE.json configs[1]): 256x256 grey X-ray-like images replicated to 3 channels, a
synthetic 10 % mask, patch_size 9.  One GPU fills a batch of B such images in lock step; a
"step" is one complete pass of the fill loop over that batch (every image inpainted from its
original mask to an empty mask), starting from inputs already resident in HBM.  With N > 1 every
rank runs its own batch on its own GPU (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0: value = inpainted images / s over all GPUs; `evals_per_s_per_gpu`
is the second half of the metric (masked patch-SSD window evaluations per second of search-kernel
time on one GPU); `roofline` prices the dominant kernel (the search); `cpu_baseline` times the
C oracle ("port") on the host cores over a bounded sample of the same images.
"""
import argparse
import json
import os
import sys
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


COLOUR = False


def _make_one(seed):
    import crim_synth
    return (crim_synth.xray_like(H, W, seed, colour=COLOUR),
            crim_synth.blob_mask(H, W, seed, fraction=MASK_FRACTION, margin=PATCH))


def make_batch(batch, rank):
    """Seeded synthetic images and masks (seed = 1000 * config + global image index), generated on
    a few host processes (forked before this process touches the GPU)."""
    import multiprocessing as mp
    seeds = [1000 * CONFIG_ID + rank * batch + i for i in range(batch)]
    imgs = np.empty((batch, H, W, 3), np.uint8)
    masks = np.empty((batch, H, W), np.uint8)
    workers = max(1, min(16, os.cpu_count() or 1, batch // 32))
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_make_one, seeds, chunksize=16)
    else:
        res = [_make_one(s) for s in seeds]
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
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    cores = max(1, min(cores, os.cpu_count() or 1))
    n = min(len(imgs), cores * budget_images_per_core)
    tasks = [(imgs[i], masks[i]) for i in range(n)]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_one, tasks)
    wall = time.time() - t0
    iters = sum(r[1] for r in res)
    return {"_crc": [r[2] for r in res],
            "value": n / wall, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d of the batch's 256x256 images (first by index), C oracle "
                      "oracle/criminisi_oracle.c, one image per process, %d processes; "
                      "%.1f s wall, %d iterations" % (n, cores, wall, iters)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--colour", action="store_true",
                    help="false-colour images (5-plane search) instead of the grey ones of BASELINE configs[1]; "
                         "side measurement, run with --no-roofline")
    args = ap.parse_args()
    global COLOUR
    COLOUR = args.colour

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    B = args.batch
    imgs, masks = make_batch(B, rank)

    # CPU baseline first (rank 0, single-GPU runs only), before this process touches the GPU:
    # the worker pool forks.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(imgs, masks)

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

    eng = capi.Engine(H, W, PATCH, batch=B, device=local_rank)

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

    if rank == 0 and not args.no_roofline:
        # Same batch once more on an engine that brackets every search launch with HIP events
        # on the engine's own streams (eager launches; the timed region above replays graphs),
        # each bracketed launch alone on the GPU: the kernel's own duration.  In the timed region
        # the two halves of the batch run on two streams and mostly take turns on the GPU, so a
        # launch there also waits for, or shares the CUs with, the other half's kernels; the
        # second pass below brackets the launches under those conditions (stream-ordered events:
        # the wait is included).
        with capi.Engine(H, W, PATCH, batch=B, device=local_rank, flags=capi.FLAG_PROFILE_SOLO) as pe:
            pe.upload_device(d_img.data_ptr(), d_mask.data_ptr())
            pe.run()
            ms, launches, evals = pe.profile()
        with capi.Engine(H, W, PATCH, batch=B, device=local_rank, flags=capi.FLAG_PROFILE) as pe:
            pe.upload_device(d_img.data_ptr(), d_mask.data_ptr())
            pe.run()
            ov_ms, ov_launches, _ = pe.profile()
        # Roofs for the search of a batch of grey images.  HBM: every iteration reads, per image,
        # the three byte planes it scores (L-128 and the two digits of Q) and the window-validity
        # bitmap once: 3 + 1/8 bytes per pixel, i.e. 3.125*H*W / ((H-8)*(W-8)) bytes per window
        # scored; the working set (gigabytes) is far beyond L2 and the 256 MB Infinity Cache.
        # MFMA: 243 algorithmic multiply-accumulates per window (960 issued, see the note).  The
        # HBM roof is the nearer one (per launch of 2048 images: 52 us against 48 us of issued
        # MFMA work, 24 us of algorithmic MFMA work), so it is the one reported.
        bytes_per_window = 3.125 * H * W / ((H - PATCH + 1) * (W - PATCH + 1))
        algo_bytes = evals * bytes_per_window
        achieved = algo_bytes / (ms * 1e-3) / 1e9
        ops = evals * PATCH * PATCH * 3 * 2
        traffic = pmc_traffic_per_launch(B / 2.0 if B >= 32 else B)
        result["evals_per_s_per_gpu"] = evals / (ms * 1e-3)
        result["roofline"] = {
            "bound": "hbm", "kernel": "k_search_pipe<9, false>",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "avg_launch_us": 1e3 * ms / max(launches, 1), "launches": launches,
            "algorithmic_bytes_per_launch": algo_bytes / max(launches, 1),
            "overlapped_avg_launch_us": 1e3 * ov_ms / max(ov_launches, 1),
            "mfma_achieved_tops": ops / (ms * 1e-3) / 1e12, "mfma_peak_tops": MFMA_I8_PEAK_TOPS,
            "mfma_frac": ops / (ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
            "note": "achieved/frac/avg_launch_us: every search launch of one pass over the batch, each "
                    "alone on the GPU; overlapped_avg_launch_us: the same with the two halves of the "
                    "batch on two streams as in the timed region (event to event, waiting for the other "
                    "half's kernels included); "
                    "algorithmic bytes = 3.125 B per pixel per image per iteration (3 byte planes + validity "
                    "bitmap), one launch covers one lane = half the batch; traffic = HBM-side bytes per "
                    "launch from the committed PMC passes (profiles/; FETCH_SIZE doubled for 16-byte "
                    "loads, plus WRITE_SIZE): tiles overlap by 8 of 40 rows, every wave load is a whole KiB (10 % padding), and the blocks of one image "
                    "share an XCD so that its L2 serves most of that overlap; mfma_*: integer ops (2 per MAC) against the dense i8 MFMA peak, 243 "
                    "algorithmic MAC per window of 960 issued (each template row occupies 32 K-slots: 16 "
                    "x shifts + 9 taps, padded; two rows per v_mfma_i32_16x16x64_i8)",
        }
    if cpu is not None:
        # the sample's outputs double as a check of the timed run's results (outside the timed region)
        import zlib
        crcs = cpu.pop("_crc")
        got, _, _ = eng.download(0, len(crcs))
        same = sum(int(zlib.crc32(got[i].tobytes()) == c) for i, c in enumerate(crcs))
        cpu["outputs_identical_to_gpu"] = "%d of %d" % (same, len(crcs))
        result["cpu_baseline"] = cpu
    eng.close()
    if rank == 0:
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
