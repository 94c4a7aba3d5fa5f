"""correlation_map_acmm at a given batch, for a kernel trace: python tools/acmm_probe.py [pairs]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import pose_estimation_amd as pea

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(1)
f1 = torch.randn(b, 128, 56, 56, device="cuda", generator=g)
f2 = torch.randn(b, 128, 56, 56, device="cuda", generator=g)
mask = (torch.rand(b, 1, 56, 56, device="cuda", generator=g) > 0.3).float()
for _ in range(10):
    out = pea.correspondence.correlation_map_acmm(f1, f2, mask)
torch.cuda.synchronize()
print(out.shape)
