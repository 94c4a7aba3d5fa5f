import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import pose_estimation_amd as pea
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for (b, c, grid, k) in ((24, 256, 128, 5), (128, 128, 64, 5), (512, 128, 64, 1)):
    x = torch.randn(2, b, c, grid * grid, device=dev, generator=g)
    x = (x / x.norm(dim=2, keepdim=True)).to(torch.bfloat16).contiguous()
    d, i, st = pea.correspondence.dense_nn(x[0], x[1], k, return_stats=True)
    print((b, c, grid, k), st, "of", b * grid * grid, flush=True)
# float32 inputs through the split-bfloat16 filter (its margin is about 30 times wider than the native filter's)
for (b, c, grid) in ((512, 128, 64), (24, 256, 128)):
    x = torch.randn(2, b, c, grid * grid, device=dev, generator=g)
    x = (x / x.norm(dim=2, keepdim=True)).contiguous()
    for flt in ("native", "split"):
        d, i, st = pea.correspondence.dense_nn(x[0], x[1], 1, return_stats=True, filter=flt)
        print((b, c, grid, 1, flt), st, "of", b * grid * grid, flush=True)
