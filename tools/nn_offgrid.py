"""dense_nn on shapes off the grid of the fast kernels (channels not a multiple of 32, cell counts not multiples
of 8), through the library's own padding (default) and with ISPA_NN_NO_PAD (guarded loaders), next to the on-grid
neighbour:  python tools/nn_offgrid.py [pairs]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import pose_estimation_amd as pea

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(5)


def timed(q, r, k, pad):
    for _ in range(2):
        pea.correspondence.dense_nn(q, r, k, pad=pad)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        pea.correspondence.dense_nn(q, r, k, pad=pad)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / 3


print(f"{b} pairs; ms per call: on-grid neighbour | off-grid, padded inside the call | off-grid, ISPA_NN_NO_PAD")
for name, c, side, cp, sidep in (("C=100 64x64", 100, 64, 128, 64), ("C=128 63x63", 128, 63, 128, 64),
                                 ("C=100 63x63", 100, 63, 128, 64), ("C=60 57x57", 60, 57, 64, 57)):
    for dtype in (torch.float32, torch.bfloat16):
        for k in (1, 5):
            def maps(ch, s):
                x = torch.randn(2, b, ch, s * s, device="cuda", generator=g)
                return (x / x.norm(dim=2, keepdim=True)).to(dtype)
            on = maps(cp, sidep if sidep % 8 == 0 else 64)
            off = maps(c, side)
            t_on = timed(on[0].contiguous(), on[1].contiguous(), k, True)
            t_pad = timed(off[0].contiguous(), off[1].contiguous(), k, True)
            t_raw = timed(off[0].contiguous(), off[1].contiguous(), k, False)
            print(f"{name:14s} {str(dtype)[6:]:9s} k={k}  {t_on:7.3f} | {t_pad:7.3f} ({t_pad / t_on:4.2f}x) | {t_raw:7.3f}", flush=True)
