"""CPU: shard arithmetic and the result all-gather on a 2-rank gloo group."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _load_sharding():
    # the package's __init__ loads the HIP library; sharding itself is pure torch
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ispa_sharding", os.path.join(here, "pose_estimation_amd", "sharding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shard_range_covers_everything():
    sh = _load_sharding()
    for total in (0, 1, 7, 8, 4096, 3600):
        for world in (1, 2, 3, 8):
            blocks = [sh.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert sh.shard_range(3600, 3, 8) == (1350, 1800)          # 450 poses per GPU
    with pytest.raises(ValueError):
        sh.shard_range(8, 8, 8)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _load_sharding()
    lo, hi = sh.shard_range(total, rank, world)
    full = torch.arange(total * 3, dtype=torch.int32).reshape(total, 3)
    got = sh.allgather_results(full[lo:hi].clone(), total)
    ok = torch.equal(got, full)
    # overlapped form: two gathers in flight, collected in order
    pend = [sh.allgather_results_async((full[lo:hi] + k).clone(), total) for k in (0, 5)]
    ok = ok and torch.equal(pend[0].wait(), full) and torch.equal(pend[1].wait(), full + 5)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def _bench_loop_worker(rank, world, port, total, q):
    """bench.py's own control flow (sharding.timed_steps) with a fake step: the rank's share of `total`
    units per step, an overlapped result gather per step, rank 1 slower than rank 0."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _load_sharding()
    lo, hi = sh.shard_range(total, rank, world)
    log = {"steps": 0, "gathered": [], "pending": []}

    def step():
        time.sleep(0.01 * (1 + 2 * rank))                          # rank 1 takes three times as long
        block = torch.full((hi - lo, 2), float(log["steps"]), dtype=torch.float32)
        block[:, 1] = torch.arange(lo, hi, dtype=torch.float32)
        for p in log["pending"]:
            log["gathered"].append(p.wait())
        log["pending"] = [sh.allgather_results_async(block, total)]
        log["steps"] += 1

    def drain():
        for p in log["pending"]:
            log["gathered"].append(p.wait())
        log["pending"] = []
    elapsed = sh.timed_steps(step, drain, steps=5, warmup=2, sync=lambda: None)
    ok = log["steps"] == 7 and len(log["gathered"]) == 7 and not log["pending"]
    for i, g in enumerate(log["gathered"]):                        # every gather complete, in unit order
        ok = ok and g.shape == (total, 2) and bool((g[:, 0] == i).all()) and torch.equal(g[:, 1], torch.arange(total, dtype=torch.float32))
    ok = ok and elapsed >= 5 * 0.03 * 0.9                          # the slowest rank sets the time on both
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, round(elapsed, 6)))


@pytest.mark.parametrize("total", [8, 7, 3600])
def test_bench_control_flow_two_ranks_gloo(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_loop_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[:2] for r in results] == [(0, True), (1, True)]
    assert results[0][2] == results[1][2]                          # MAX over ranks: the same number everywhere


@pytest.mark.parametrize("total", [8, 7])
def test_allgather_two_ranks_gloo(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


@pytest.mark.gpu
def test_allgather_device_tensors_one_rank_rccl(tmp_path):
    """The exchange as the GPU job runs it: device tensors through RCCL (one rank is all a one-GPU
    box can hold), synchronous and overlapped forms, in this process."""
    sh = _load_sharding()
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        full = torch.arange(512 * 4096, dtype=torch.int32, device="cuda").reshape(512, 1, 4096)
        assert torch.equal(sh.allgather_results(full.clone(), 512), full)
        pend = [sh.allgather_results_async(full + k, 512) for k in (0, 7)]
        busy = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")   # work meanwhile
        assert torch.equal(pend[0].wait(), full) and torch.equal(pend[1].wait(), full + 7)
        assert torch.isfinite(busy).all()
        with pytest.raises(ValueError):
            sh.allgather_results(full[:100], 512)
    finally:
        dist.destroy_process_group()
