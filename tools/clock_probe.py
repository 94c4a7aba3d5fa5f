#!/usr/bin/env python3
"""Filter-kernel time of the bfloat16 dense NN on random features against all-zero and constant features.

The instruction stream is the same for every input; what changes is how many bits toggle in the matrix and
vector pipes, i.e. the power the loop draws and the clock the chip holds under it.  A large gap between the
random and the zero timing says the kernel is limited by the clock the chip sustains, not by instruction issue."""
import ctypes
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea
from pose_estimation_amd import _abi


def filt_ms(q, r, k, steps=20):
    for _ in range(3):
        pea.correspondence.dense_nn(q, r, k)
    _abi.check(_abi.lib.ispa_timing_enable(steps))
    for _ in range(steps):
        pea.correspondence.dense_nn(q, r, k)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * steps)()
    cnt = ctypes.c_int(0)
    _abi.check(_abi.lib.ispa_timing_collect(ms, steps, ctypes.byref(cnt)))
    _abi.check(_abi.lib.ispa_timing_enable(0))
    return float(np.median(ms[:cnt.value]))


def main():
    b, c, g = 512, 128, 64
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(b, c, g, g, device=dev, generator=gen)
    rnd = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()
    y = torch.randn(b, c, g, g, device=dev, generator=gen)
    rnd2 = (y / y.norm(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()
    zero = torch.zeros_like(rnd)
    const = torch.full_like(rnd, 0.0884)
    flop = 2.0 * (g * g) ** 2 * c * b
    for name, q, r in (("random", rnd, rnd2), ("zeros", zero, zero), ("constant", const, const), ("random", rnd, rnd2)):
        ms = filt_ms(q, r, k)
        print(f"k={k} {name:9s} filter {ms:.3f} ms  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)


main()
