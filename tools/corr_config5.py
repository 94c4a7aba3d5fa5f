"""Timing probe: eccv keypoint correlation maps at the config-5 size (3 views of 128x128x256
features, K keypoints per view, no resize) through CorrelationMapLayer.forward_views.
    python tools/corr_config5.py [K] [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda").manual_seed(1)
f1 = torch.randn(b, 3, 256, 128, 128, device="cuda", generator=g)
f2 = torch.randn(b, 3, 256, 128, 128, device="cuda", generator=g)
f1 /= f1.norm(dim=2, keepdim=True)
f2 /= f2.norm(dim=2, keepdim=True)
knn = np.random.RandomState(1).randint(0, 128, (3, k, 2))
layer = pea.correspondence.CorrelationMapLayer(grid=128)
for _ in range(3):
    layer.forward_views(f1, f2, knn)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
steps = 20
ev[0].record()
for _ in range(steps):
    layer.forward_views(f1, f2, knn)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / steps
print(f"K={k} batch={b}: {ms:.3f} ms per call ({b / ms * 1e3:.0f} queries/s, {3 * b * k / ms * 1e3:.0f} keypoint maps/s)")
