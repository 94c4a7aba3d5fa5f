"""One-off randomized differential check of the correlation maps (eccv per-view call, acmm) against the numpy
oracle (development tool, needs a GPU): random batch / channel / grid sizes and keypoint counts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea                       # noqa: E402
from oracle import correlation as ocorr                 # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    rs = np.random.RandomState(seed)
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        b, c = int(rs.randint(1, 7)), int(rs.choice([1, 3, 16, 31, 64, 128, 130]))
        grid = int(rs.choice([4, 7, 14, 28, 28, 30, 46]))
        hw = int(rs.choice([grid, 2 * grid, 2 * grid - 1]))
        f1 = rs.randn(b, c, hw, hw).astype(np.float32)
        f2 = rs.randn(b, c, hw, hw).astype(np.float32)
        f1 /= np.sqrt((f1 ** 2).sum(1, keepdims=True))
        f2 /= np.sqrt((f2 ** 2).sum(1, keepdims=True))
        k = int(rs.randint(1, 21))
        knn = rs.randint(0, grid, (k, 2))
        layer = pea.correspondence.CorrelationMapLayer(grid)
        got = layer.forward(f1, f2, knn)
        want = ocorr.correlation_map_eccv(f1, f2, knn, grid)
        ok = np.allclose(got, want, rtol=1e-4, atol=1e-6)
        why = "eccv"
        if ok and (hw + 1) // 2 == grid:
            mask = (rs.rand(b, 1, hw, hw) > 0.3).astype(np.float32)
            ga = pea.correspondence.correlation_map_acmm(f1, f2, mask, grid)
            wa = ocorr.correlation_map_acmm(f1, f2, mask, grid)
            ok = np.array_equal(np.isnan(ga), np.isnan(wa)) and np.allclose(ga, wa, rtol=1e-4, atol=1e-6, equal_nan=True)
            why = "acmm"
        cases += 1
        if not ok:
            print(f"FAIL case {cases} ({why}): b={b} c={c} grid={grid} hw={hw} k={k} seed={seed}", flush=True)
            sys.exit(1)
        if cases % 10 == 0:
            print(f"{cases} cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"all {cases} cases ok")


if __name__ == "__main__":
    main()
