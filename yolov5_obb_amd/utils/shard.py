"""Image sharding for multi-GPU inference / validation: one process per GPU, images split by
rank, NO collective on the data path; the only exchange is the final gather of the (small) per-image results on
rank 0, where the reference computes its metrics (val.py:269-274).  Works with any torch.distributed backend
(RCCL = "nccl" on MI355X nodes, "gloo" in the CPU tests).

The reference's val.py / detect.py are single-process (utils/torch_utils.py:83 returns cuda:0); this is the thin layer
that lets N copies of them split one image list.
"""
import torch
import torch.distributed as dist


def world():
    """(rank, world_size) of the default process group, (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n, rank=None, world_size=None):
    """Indices of the items rank `rank` processes: a strided split (rank, rank+world, ...) so that every rank gets
    a similar mix of the list (DOTA tiles of one source image are adjacent and similarly dense)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n, world_size))


def gather_results(local_indices, local_results, n_total, dst=0):
    """Collect per-item results on rank `dst` in the original item order.

    local_indices: the indices returned by shard_indices; local_results: one picklable object per index (e.g. the
    (n,7) detection array of an image, or the (correct, conf, pcls, tcls) stats tuple of val.py:250).
    Returns the full list (length n_total) on rank dst, None elsewhere.  Single-process: returns the local list."""
    if len(local_indices) != len(local_results):
        raise ValueError("one result per local index expected")
    rank, ws = world()
    if ws == 1:
        out = [None] * n_total
        for i, r in zip(local_indices, local_results):
            out[i] = r
        return out
    payload = (list(local_indices), [r.cpu() if isinstance(r, torch.Tensor) else r for r in local_results])
    gathered = [None] * ws if rank == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if rank != dst:
        return None
    out = [None] * n_total
    seen = 0
    for idx, res in gathered:
        for i, r in zip(idx, res):
            if out[i] is not None:
                raise RuntimeError(f"item {i} produced by two ranks")
            out[i] = r
            seen += 1
    if seen != n_total:
        raise RuntimeError(f"{n_total - seen} items missing after the gather")
    return out


def max_over_ranks(seconds, device=None):
    """MAX over ranks of a local wall time (the bench / speed-test convention: the job is as slow as its slowest rank)."""
    rank, ws = world()
    if ws == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
