"""world_size-2 test of the per-image sharding path on CPU (gloo).  The GPU engine is replaced by
the C oracle as the per-rank worker: what is under test is the partition / gather logic that
bench.py --gpus N and the batch driver rely on, not the arithmetic."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from helpers import ROOT, PKG


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_worker(images, masks, patch_size):
    from oracle import c_oracle
    out = np.empty_like(images)
    for i in range(len(images)):
        out[i] = c_oracle.run(images[i], masks[i], patch_size, want_trace=False)[0]
    return out


def _rank_main(rank, world, port, n, tmpdir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import crim_shard
    import crim_synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    imgs = np.stack([crim_synth.xray_like(40, 44, 50 + i, colour=(i % 2 == 0)) for i in range(n)])
    masks = np.stack([crim_synth.blob_mask(40, 44, 50 + i, fraction=0.05, margin=8) for i in range(n)])
    assert crim_shard.shard_indices(n, world, rank) == list(range(rank, n, world))
    out = crim_shard.inpaint_sharded(imgs, masks, 7, _oracle_worker, dist)
    np.save(os.path.join(tmpdir, "out%d.npy" % rank), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_serial(tmp_path):
    n, world = 5, 2
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, PKG)
    import crim_synth
    imgs = np.stack([crim_synth.xray_like(40, 44, 50 + i, colour=(i % 2 == 0)) for i in range(n)])
    masks = np.stack([crim_synth.blob_mask(40, 44, 50 + i, fraction=0.05, margin=8) for i in range(n)])
    serial = _oracle_worker(imgs, masks, 7)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "out%d.npy" % r))
        assert np.array_equal(got, serial)


def test_single_process_path():
    sys.path.insert(0, PKG)
    import crim_shard
    import crim_synth
    imgs = np.stack([crim_synth.xray_like(32, 32, 70 + i) for i in range(3)])
    masks = np.stack([crim_synth.blob_mask(32, 32, 70 + i, fraction=0.05, margin=8) for i in range(3)])
    out = crim_shard.inpaint_sharded(imgs, masks, 5, _oracle_worker, None)
    assert np.array_equal(out, _oracle_worker(imgs, masks, 5))
