"""Sharding of independent work units (image pairs, poses) over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  Units share nothing,
so the data path has no collective; the only exchange is one all-gather of the fixed-size
result records at the end of a batch.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world_size):
    """Contiguous block [lo, hi) of `total` units owned by `rank`; blocks differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside the world")
    base, extra = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(total, world_size):
    return [shard_range(total, r, world_size)[1] - shard_range(total, r, world_size)[0]
            for r in range(world_size)]


def allgather_results(local, total, group=None):
    """Gather per-rank result blocks (first axis = units of this rank) into the full batch.

    `local` is this rank's (n_local, ...) tensor; every rank returns the (total, ...) tensor in
    unit order.  Uneven shards are padded to the largest block for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != total:
            raise ValueError("no process group: the local block must be the whole batch")
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match this rank's shard")
    biggest = max(sizes)
    if all(s == biggest for s in sizes):
        out = torch.empty((world * biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


class PendingGather:
    """Result all-gather in flight (even shards): `wait()` returns the (total, ...) tensor.

    The collective runs on the communicator's own stream, so kernels of the next batch launched
    meanwhile overlap with it; the object keeps the input block alive until then."""

    def __init__(self, work, out, local):
        self._work, self._out, self._local = work, out, local

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = self._local = None
        return self._out


def allgather_results_async(local, total, group=None):
    """Start the all-gather of this rank's result block and return a PendingGather.

    Without a process group, or with uneven shards, the gather is done at once (the object is
    already complete)."""
    if not (dist.is_available() and dist.is_initialized()):
        return PendingGather(None, allgather_results(local, total, group), None)
    world = dist.get_world_size(group)
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match this rank's shard")
    if any(s != sizes[0] for s in sizes):
        return PendingGather(None, allgather_results(local, total, group), None)
    local = local.contiguous()
    out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local, group=group, async_op=True)
    return PendingGather(work, out, local)
