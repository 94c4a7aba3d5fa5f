"""Throughput of the correlation maps over batch sizes (reference shapes: 128x56x56 features -> 28x28 maps).
python tools/corr_sweep.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

import pose_estimation_amd as pea


def timed(fn, reps=5):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


g = torch.Generator(device="cuda").manual_seed(1)
layer = pea.correspondence.CorrelationMapLayer()
print("  B    K   eccv ms  GB/s(out)   acmm ms  TFLOP/s")
for b in (1, 8, 64, 256, 1024):
    f1 = torch.randn(b, 128, 56, 56, device="cuda", generator=g)
    f2 = torch.randn(b, 128, 56, 56, device="cuda", generator=g)
    mask = (torch.rand(b, 1, 56, 56, device="cuda", generator=g) > 0.3).float()
    for k in (1, 10, 14):
        knn = np.random.RandomState(1).randint(0, 28, (k, 2))
        ms_e = timed(lambda: layer.forward(f1, f2, knn))
        ms_a = timed(lambda: pea.correspondence.correlation_map_acmm(f1, f2, mask)) if k == 10 else float("nan")
        print(f"{b:5d} {k:3d} {ms_e:9.3f} {b * k * 784 * 4 / ms_e / 1e6:9.1f} {ms_a:9.3f} "
              f"{2.0 * b * 784 * 784 * 128 / ms_a / 1e9:8.1f}", flush=True)
