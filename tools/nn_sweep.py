"""Whole-call throughput of dense_nn over channel counts, k and input types: a table to spot performance cliffs.
python tools/nn_sweep.py [pairs]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import pose_estimation_amd as pea

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(5)
print(f"{b} pairs of 64x64 maps; TFLOP/s of 2 nq nr C per pair over the whole call")
print("   C   k " + " ".join(f"{m:>12s}" for m in ("f32", "bf16", "f32-split")))
for c in (32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 384, 512, 100):
    x = torch.randn(2, b, c, 4096, device="cuda", generator=g)
    x = x / x.norm(dim=2, keepdim=True)
    for k in (1, 5, 8):
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
            cells.append(f"{2.0 * b * 4096 * 4096 * c / (ms * 1e-3) / 1e12:7.1f} {ms:6.2f}ms"[:14])
        print(f"{c:4d} {k:3d} " + " ".join(f"{x_:>14s}" for x_ in cells), flush=True)
