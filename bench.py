#!/usr/bin/env python3
"""Throughput bench of the dense correspondence stage (BASELINE.json's metric).

A step = one pass of the hot path over one batch of synthetic image pairs resident in HBM:
dense nearest-neighbour correspondence between a 64x64 grid of 128-d query features and a
64x64 grid of template features per pair (ispa_nn_topk through the C ABI: norms, MFMA filter,
finalize, float64 exact pass).  With N GPUs each rank holds its own batch (weak scaling, no
collective in the data path) and the index/distance records are all-gathered once per step.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}       # dense peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=512, help="image pairs per GPU per step")
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--no-gather", action="store_true", help="skip the per-step result all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(grid, channels, k, budget_s):
    """numpy float32 formulation (sgemm + |a|^2+|b|^2-2ab + argmin) on the host cores."""
    from oracle import nn as onn
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    rs = np.random.RandomState(2)
    n = grid * grid
    q = rs.randn(channels, n).astype(np.float32)
    r = rs.randn(channels, n).astype(np.float32)
    q /= np.sqrt((q ** 2).sum(0, keepdims=True))
    r /= np.sqrt((r ** 2).sum(0, keepdims=True))
    onn.nn_sgemm(q, r, k)                                   # warm
    done, t0 = 0, time.perf_counter()
    while True:
        onn.nn_sgemm(q, r, k)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 400:
            break
    return {"value": round(done / el, 3), "unit": "pairs/s", "cores": int(cores), "kind": "port",
            "sample": f"{done} pairs of {grid}x{grid}x{channels} float32 in {el:.1f} s, numpy sgemm + argmin "
                      f"(oracle.nn.nn_sgemm), BLAS threads = {cores}"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    import pose_estimation_amd as pea           # raises if libispa_hip.so is missing
    from pose_estimation_amd import _abi, sharding

    dev = torch.device("cuda", torch.cuda.current_device())
    b, c, g = args.pairs, args.channels, args.grid
    n = g * g
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    def feats():
        x = torch.randn(b, c, g, g, device=dev, generator=gen)
        return (x / x.norm(dim=1, keepdim=True)).to(tdt).contiguous()
    q, r = feats(), feats()
    gather = world > 1 and not args.no_gather

    def step():
        dist_, ind_ = pea.correspondence.dense_nn(q, r, args.k)
        if gather:
            sharding.allgather_results(ind_, b * world)
            sharding.allgather_results(dist_, b * world)
        return dist_, ind_

    for _ in range(args.warmup):
        step()
    _abi.check(_abi.lib.ispa_timing_enable(args.steps))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-launch duration of the filter kernel, HIP events on the launch stream inside the timed region
    ms = (ctypes.c_float * args.steps)()
    cnt = ctypes.c_int(0)
    _abi.check(_abi.lib.ispa_timing_collect(ms, args.steps, ctypes.byref(cnt)))
    _abi.check(_abi.lib.ispa_timing_enable(0))
    filt_ms = float(np.mean(ms[:cnt.value])) if cnt.value else float("nan")

    if rank == 0:
        pairs_total = b * world * args.steps
        flop_per_pair = 2.0 * n * n * c                          # GEMM term of the distance stage
        achieved = flop_per_pair * b / (filt_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        out = {
            "metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"],
            "value": round(pairs_total / elapsed, 2),
            "unit": "maps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"dense NN correspondence, {g}x{g} grid x {c}-d features, k={args.k}, "
                                   f"{b} pairs per GPU per step ({args.dtype} MFMA filter + float64 exact pass)",
                       "pairs_per_gpu": b, "grid": g, "channels": c, "k": args.k,
                       "result_allgather": bool(gather)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None,
                         "kernel": "nn_filter", "kernel_ms": round(filt_ms, 4),
                         "flop_per_launch": flop_per_pair * b},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(g, c, args.k, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
