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


def _padded_block(local, biggest):
    if local.shape[0] == biggest:
        return local.contiguous()
    padded = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    return padded


def _unpad(out, sizes, biggest):
    """(world * biggest, ...) gathered blocks -> (total, ...) in unit order."""
    if all(s == biggest for s in sizes):
        return out
    return torch.cat([out[r * biggest:r * biggest + s] for r, s in enumerate(sizes)], dim=0)


def allgather_results(local, total, group=None):
    """Gather per-rank result blocks (first axis = units of this rank) into the full batch.

    `local` is this rank's (n_local, ...) tensor; every rank returns the (total, ...) tensor in
    unit order.  One all_gather_into_tensor either way: uneven shards (they differ by at most one
    unit) are padded to the largest block and the padding is cut out afterwards."""
    return allgather_results_async(local, total, group).wait()


class PendingGather:
    """Result all-gather in flight: `wait()` returns the (total, ...) tensor.

    The collective runs on the communicator's own stream, so kernels of the next batch launched
    meanwhile overlap with it; the object keeps the input block alive until then."""

    def __init__(self, work, out, local, sizes=None, biggest=0):
        self._work, self._out, self._local, self._sizes, self._biggest = work, out, local, sizes, biggest

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = self._local = None
            if self._sizes is not None:
                self._out = _unpad(self._out, self._sizes, self._biggest)
                self._sizes = None
        return self._out


def allgather_results_async(local, total, group=None):
    """Start the all-gather of this rank's result block and return a PendingGather.

    Without a process group the object is already complete (the local block must be the whole batch)."""
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != total:
            raise ValueError("no process group: the local block must be the whole batch")
        return PendingGather(None, local, None)
    world = dist.get_world_size(group)
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match this rank's shard")
    biggest = max(sizes)
    block = _padded_block(local, biggest)
    out = torch.empty((world * biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, block, group=group, async_op=True)
    return PendingGather(work, out, block, sizes, biggest)


def timed_steps(step, drain, steps, warmup, sync, group=None):
    """The measurement loop of bench.py, shared by every workload and rank: `warmup` untimed steps, then
    EXACTLY `steps` timed ones between two (barrier + sync()) pairs; `drain()` completes whatever a step left
    in flight (result gathers) and runs inside the timed region, before the closing barrier.  Returns the
    elapsed seconds of the slowest rank (MAX over ranks), the same value on every rank.  `sync` waits for
    the device (torch.cuda.synchronize; a no-op for host-only steps)."""
    import time
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    for _ in range(warmup):
        step()
    drain()
    if multi:
        dist.barrier(group=group)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    if multi:
        dist.barrier(group=group)
    sync()
    elapsed = time.perf_counter() - t0
    if multi:
        backend = dist.get_backend(group)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    return elapsed
