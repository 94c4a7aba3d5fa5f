"""dense_nn at the reference's own shape (ucn_code/nets/train.prototxt:2116-2141): a few hundred sampled query
features per image against all 56x56 cells of the other image, small batches.  python tools/nn_sweep_queries.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import pose_estimation_amd as pea

g = torch.Generator(device="cuda").manual_seed(5)
print("    B    nq    nr   k          f32 ms      bf16 ms")
for b, nq, nr in ((1, 1000, 3136), (8, 1000, 3136), (8, 999, 3136), (8, 1000, 3025), (64, 1000, 3136), (8, 200, 3136),
                  (8, 1000, 784), (512, 1000, 3136), (8, 3136, 3136)):
    q = torch.randn(b, 128, nq, device="cuda", generator=g)
    r = torch.randn(b, 128, nr, device="cuda", generator=g)
    q, r = q / q.norm(dim=1, keepdim=True), r / r.norm(dim=1, keepdim=True)
    for k in (1, 5):
        cells = []
        for dtype in (torch.float32, torch.bfloat16):
            qq, rr = q.to(dtype).contiguous(), r.to(dtype).contiguous()
            for _ in range(3):
                pea.correspondence.dense_nn(qq, rr, k)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5):
                pea.correspondence.dense_nn(qq, rr, k)
            t1.record()
            torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 5
            cells.append(f"{ms:7.3f} ({2.0 * b * nq * nr * 128 / ms / 1e9:6.1f} TF)")
        print(f"{b:5d} {nq:5d} {nr:5d} {k:3d}   " + "  ".join(cells), flush=True)
