"""Latency of one dense-NN call on a single 64x64x128 pair: host enqueue cost, pipelined rate, synchronous latency."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import pose_estimation_amd as pea


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(7)
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for tdt in (torch.float32, torch.bfloat16):
        x = torch.randn(2, pairs, 128, 64, 64, device=dev, generator=gen)
        x = (x / x.norm(dim=2, keepdim=True)).to(tdt).contiguous()
        q, r = x[0], x[1]
        fn = lambda: pea.correspondence.dense_nn(q, r, 1)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n = 300
        host = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            a = time.perf_counter()
            fn()
            host += time.perf_counter() - a
            torch.cuda.synchronize()
        sync_lat = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / n
        print(f"{tdt}: host enqueue {host / n * 1e6:.1f} us, pipelined {piped * 1e6:.1f} us/call, "
              f"synchronous {sync_lat * 1e6:.1f} us/call", flush=True)


if __name__ == "__main__":
    main()
