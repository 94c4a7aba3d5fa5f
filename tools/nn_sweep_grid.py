"""Whole-call throughput of dense_nn over map sizes (C = 128): the reference's 56x56 and 28x28 lattices, odd sizes.
python tools/nn_sweep_grid.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import pose_estimation_amd as pea

g = torch.Generator(device="cuda").manual_seed(5)
c = 128
print("grid pairs   k " + " ".join(f"{m:>14s}" for m in ("f32", "bf16", "f32-split")) + "   (TFLOP/s, ms per call)")
for grid, b in ((28, 512), (56, 128), (56, 8), (60, 64), (64, 64), (96, 32), (128, 8), (128, 1), (63, 64), (32, 1), (64, 1)):
    n = grid * grid
    x = torch.randn(2, b, c, n, device="cuda", generator=g)
    x = x / x.norm(dim=2, keepdim=True)
    for k in (1, 5):
        cells = []
        for dtype, flt in ((torch.float32, "native"), (torch.bfloat16, "native"), (torch.float32, "split")):
            if flt == "split" and k != 1:
                cells.append("-")
                continue
            q, r = x[0].to(dtype).contiguous(), x[1].to(dtype).contiguous()
            for _ in range(2):
                pea.correspondence.dense_nn(q, r, k, filter=flt)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(3):
                pea.correspondence.dense_nn(q, r, k, filter=flt)
            t1.record()
            torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 3
            cells.append(f"{2.0 * b * n * n * c / (ms * 1e-3) / 1e12:7.1f} {ms:6.3f}")
        print(f"{grid:4d} {b:5d} {k:3d} " + " ".join(f"{x_:>14s}" for x_ in cells), flush=True)
