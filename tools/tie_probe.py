"""Time dense_nn on maps with large exactly-tied regions (masked / zero-padded cells): every query of such a region
goes through the float64 pass.  python tools/tie_probe.py [pairs]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import pose_estimation_amd as pea

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
for frac in (0.0, 0.05, 0.5):
    x = torch.randn(2, b, 128, 4096, device=dev, generator=g)
    x = x / x.norm(dim=2, keepdim=True)
    nz = int(4096 * frac)
    if nz:
        x[:, :, :, :nz] = 0.0
    for dt, flt in ((torch.float32, "native"), (torch.float32, "split"), (torch.bfloat16, "native")):
        q, r = x[0].to(dt).contiguous(), x[1].to(dt).contiguous()
        for _ in range(2):
            d, i, st = pea.correspondence.dense_nn(q, r, 1, return_stats=True, filter=flt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            d, i = pea.correspondence.dense_nn(q, r, 1, filter=flt)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        ok = bool((i[:, 0, :nz] == 1).all()) if nz else True     # zero queries: first zero reference, 1-based
        print(f"zero cells {frac:4.2f}  {str(dt)[6:]:9s} {flt:7s} {ms:9.3f} ms/call  {st}  ties->lowest index: {ok}", flush=True)
