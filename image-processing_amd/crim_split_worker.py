#!/usr/bin/env python3
"""One rank of the intra-image split search (BASELINE configs[4]: one 4096x4096 image, the window
origins of every iteration's search divided among the GPUs of a node, one RCCL all-gather of 16
bytes per rank per iteration), run as a process of its own.

The process that drives RCCL through libcriminisi_hip.so must not also hold PyTorch's bundled ROCm
runtime, so a launcher that lives in a PyTorch process (bench.py under torch.distributed.run)
starts one of these per rank BEFORE it touches the GPU itself, lets them prepare (synthesise the
image) and releases them later through a file.  Rendezvous is a directory on the node:

    ready.<rank>   written by each worker once its inputs are prepared
    go             created by the launcher of rank 0 to start the measurement
    id             the 128-byte RCCL unique id, written by rank 0's worker
    result.<rank>  one JSON object per worker

    python crim_split_worker.py --rank R --world N --device D --rendezvous DIR
                                [--size 4096] [--patch 9] [--iters 2000] [--dry-run]

--dry-run exchanges the id and reports without touching a GPU (plumbing test on CPU).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ---- rendezvous helpers (shared with the launcher) ----------------------------------------------

def write_atomic(path, data):
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def wait_for_file(path, timeout_s, poll_s=0.05):
    """Contents of `path` once it exists; None after timeout_s seconds."""
    deadline = time.time() + timeout_s
    while True:
        if os.path.exists(path):
            with open(path, "rb") as f:
                return f.read()
        if time.time() >= deadline:
            return None
        time.sleep(poll_s)


def launch_worker(rank, world, device, rendezvous, size=4096, patch=9, iters=2000, dry_run=False,
                  python=None, transport="rccl", dump_trace=False, wait=900.0):
    """Starts this module as a child process for one rank; returns the Popen object (its log file is
    closed in this process once the child holds it)."""
    os.makedirs(rendezvous, exist_ok=True)
    cmd = [python or sys.executable, os.path.abspath(__file__), "--rank", str(rank), "--world", str(world),
           "--device", str(device), "--rendezvous", rendezvous, "--size", str(size), "--patch", str(patch),
           "--iters", str(iters), "--transport", transport, "--wait", str(wait)]
    if dry_run:
        cmd.append("--dry-run")
    if dump_trace:
        cmd.append("--dump-trace")
    with open(os.path.join(rendezvous, "worker.%d.log" % rank), "wb") as log:
        return subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)


def release(rendezvous):
    write_atomic(os.path.join(rendezvous, "go"), b"go")


def collect(rendezvous, world, timeout_s):
    """Result objects of all ranks (None for a rank that did not report in time)."""
    out = []
    deadline = time.time() + timeout_s
    for r in range(world):
        data = wait_for_file(os.path.join(rendezvous, "result.%d" % r), max(0.0, deadline - time.time()))
        out.append(json.loads(data.decode()) if data else None)
    return out


def summarise(results):
    """One object for the benchmark line from the per-rank results."""
    ok = [r for r in results if r]
    if len(ok) != len(results) or not ok:
        return {"error": "%d of %d split workers reported" % (len(ok), len(results)),
                "errors": [r.get("error") for r in ok if r.get("error")]}
    errs = [r["error"] for r in ok if r.get("error")]
    if errs:
        return {"error": errs[0]}
    r0 = ok[0]
    agree = len(set(r["trace_signature"] for r in ok)) == 1
    out = dict(r0)
    out["us_per_iteration"] = max(r["us_per_iteration"] for r in ok)
    out["all_ranks_agree_on_the_trace"] = agree
    out["rccl_ranks_seen"] = [r.get("rccl_ranks") for r in ok]
    return out


# ---- the worker ---------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--rendezvous", required=True)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--patch", type=int, default=9)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--wait", type=float, default=900.0, help="seconds to wait for the go file")
    ap.add_argument("--transport", choices=("rccl", "host"), default="rccl",
                    help="how the ranks exchange their records: RCCL all-gather, or host memory shared by the "
                         "ranks' processes (tests; ranks that share one GPU)")
    ap.add_argument("--dump-trace", action="store_true",
                    help="save this rank's per-iteration records of the split run as trace.<rank>.npy")
    args = ap.parse_args()
    rdv, rank, world = args.rendezvous, args.rank, args.world
    res = {"rank": rank, "world": world, "size": args.size, "patch": args.patch, "iterations": args.iters}

    def finish(code=0):
        write_atomic(os.path.join(rdv, "result.%d" % rank), json.dumps(res).encode())
        sys.exit(code)

    try:
        if not args.dry_run:
            sys.path[:0] = [ROOT, HERE]
            import numpy as np
            import crim_hip_capi as capi
            import crim_synth
            S = args.size
            img = crim_synth.xray_like(S, S, 4000 + S, colour=True)
            mask = crim_synth.square_hole_mask(S, S, S // 2)
        write_atomic(os.path.join(rdv, "ready.%d" % rank), b"ready")
        if wait_for_file(os.path.join(rdv, "go"), args.wait) is None:
            res["error"] = "rank %d: no go file within %.0f s" % (rank, args.wait)
            finish(3)
        # id exchange
        idpath = os.path.join(rdv, "id")
        if rank == 0:
            uid = os.urandom(128) if (args.dry_run or args.transport == "host") else capi.split_unique_id()
            write_atomic(idpath, uid)
        else:
            uid = wait_for_file(idpath, 120.0)
            if uid is None or len(uid) != 128:
                res["error"] = "rank %d: no RCCL id within 120 s" % rank
                finish(4)
        res["id_sha1"] = hashlib.sha1(uid).hexdigest()
        if args.dry_run:
            res["trace_signature"] = 0
            res["us_per_iteration"] = 0.0
            res["rccl_ranks"] = world
            finish(0)

        def timed(eng, dump=None):
            eng.upload(img, mask)
            eng.run(max_iters=32)
            t = time.perf_counter()
            eng.run(max_iters=args.iters)
            dt = time.perf_counter() - t
            tr = eng.trace(0)
            if dump:
                np.save(dump, tr)
            sig = int(np.bitwise_xor.reduce(tr["sy"].astype(np.int64) * 65536 + tr["sx"]))
            return 1e6 * dt / args.iters, sig

        with capi.Engine(S, S, args.patch, device=args.device) as eng:
            if rank == 0:
                res["single_gpu_us_per_iteration"], res["single_gpu_trace_signature"] = timed(eng)
            res["transport"] = args.transport
            if args.transport == "host":
                eng.split_init_host(world, rank, os.path.join(rdv, "exchange." + hashlib.sha1(uid).hexdigest()[:12]))
                res["rccl_ranks"] = None
            else:
                eng.split_init(world, rank, uid)
                n_seen, r_seen = eng.split_info()
                res["rccl_ranks"] = n_seen
                if n_seen != world or r_seen != rank:
                    res["error"] = "RCCL reports rank %d of %d, expected %d of %d" % (r_seen, n_seen, rank, world)
                    finish(5)
            res["us_per_iteration"], res["trace_signature"] = timed(
                eng, os.path.join(rdv, "trace.%d.npy" % rank) if args.dump_trace else None)
            res["graph_captured"], res["host_exchanges"] = eng.split_status()
        if rank == 0:
            res["split_matches_single_gpu"] = res["trace_signature"] == res["single_gpu_trace_signature"]
        finish(0)
    except SystemExit:
        raise
    except Exception as e:                                  # noqa: BLE001 (reported through the result file)
        res["error"] = "rank %d: %s: %s" % (rank, type(e).__name__, e)
        finish(1)


if __name__ == "__main__":
    main()
