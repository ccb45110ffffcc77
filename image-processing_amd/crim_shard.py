"""Per-image sharding of a batch over the GPUs of one node (one process per GPU).

Images are independent (the per-file loop of /root/reference/scripts/main.py:44-58 carries no
state from one image to the next), so image i goes to rank i mod world_size and the data path
needs no collective; only the results are gathered.  The worker is injected so that the same
code runs on the HIP engine (production) and on a stand-in (CPU tests of the sharding logic).
"""
import numpy as np


def shard_indices(n_images, world_size, rank):
    """Indices of the images rank `rank` owns: i mod world_size == rank."""
    return list(range(rank, n_images, world_size))


def inpaint_sharded(images, masks, patch_size, worker, dist=None):
    """Every rank calls this with the same full `images` [n,H,W,3] / `masks` [n,H,W].

    worker(images_subset, masks_subset, patch_size) -> uint8 [m,H,W,3] runs on this rank's
    shard.  Returns the full result array on every rank (all_gather of the shards), in the
    original image order.  With dist=None (or a single rank) everything runs locally.
    """
    images = np.asarray(images)
    masks = np.asarray(masks)
    n = images.shape[0]
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = shard_indices(n, world, rank)
    local = worker(images[mine], masks[mine], patch_size) if mine else \
        np.zeros((0,) + images.shape[1:], np.uint8)
    if world == 1:
        out = np.empty_like(images)
        out[mine] = local
        return out
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, np.ascontiguousarray(local)))
    out = np.empty_like(images)
    seen = 0
    for idx, arr in gathered:
        if len(idx):
            out[idx] = arr
            seen += len(idx)
    if seen != n:
        raise RuntimeError("sharded inpaint: gathered %d of %d images" % (seen, n))
    return out


def hip_worker(device=0, batch=None):
    """Worker running this rank's shard on the HIP engine of GPU `device`."""
    def run(images, masks, patch_size):
        from crim_hip_inpainter import inpaint_many
        return inpaint_many(images, masks, patch_size=patch_size, device=device, batch=batch)
    return run
