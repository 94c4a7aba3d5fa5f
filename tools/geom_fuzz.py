"""One-off randomized differential check of project_prune (both visibility variants, with and without crop) against
the CPU oracle (development tool, needs a GPU): random models, poses incl. close cameras, depth maps, crops, odd
point counts (scalar kernels) and shared / per-pose depth maps."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea                       # noqa: E402
from oracle import cnative                              # noqa: E402
from oracle import geometry as og                       # noqa: E402
from oracle import zbuffer as ozb                       # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    rs = np.random.RandomState(seed)
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        n = int(rs.randint(1, 12000))
        pts = rs.uniform(-0.4, 0.4, (n, 3))
        if n > 4:
            pts[rs.randint(0, n)] *= 40.0
        m = int(rs.randint(1, 20))
        params = np.stack([rs.uniform(0, 360, m), rs.uniform(-10, 40, m), rs.uniform(-10, 10, m),
                           rs.choice([0.6, 1.5, 3.0], m)], axis=1)
        cams = pea.pt2render.camera_matrices(params)
        shared = rs.rand() < 0.4
        depths = []
        for p in range(1 if shared else m):
            r = cnative.project_fma(pts, cams[p])
            with np.errstate(invalid="ignore", over="ignore"):
                d = ozb.zbuffer(r, 540, 960, radius=int(rs.randint(0, 3)))[2]
            d[~np.isfinite(d)] = float(rs.choice([3.0, 20.0]))
            depths.append(d.astype(np.float32))
        depth_arg = depths[0] if shared else np.stack(depths)
        variant = "A" if rs.rand() < 0.5 else "B"
        crop = None
        if rs.rand() < 0.7:
            y0, x0 = rs.randint(-20, 300, m), rs.randint(-20, 600, m)
            crop = np.stack([y0, y0 + rs.randint(2, 400, m), x0, x0 + rs.randint(2, 500, m)], axis=1)
            if rs.rand() < 0.3:
                crop = crop[:1]
        truth, out, td, tc = pea.run_proj.project_prune(pts, cams, depth_arg, crop, variant=variant, want_parts=True)
        ok = True
        for p in range(m):
            r = cnative.project_fma(pts, cams[p])
            d = depths[0] if shared else depths[p]
            with np.errstate(invalid="ignore", over="ignore"):
                if variant == "A":
                    dt, _ = og.depth_visibility_array_a(r, d)
                else:
                    dt, _ = og.depth_visibility_array_b(r, d)
                want = r.copy()
                ct = np.ones(n)
                if crop is not None:
                    ct, want = og.crop_visibility_array(want, [int(v) for v in crop[p if len(crop) > 1 else 0]])
            same = np.array_equal(out[p], want, equal_nan=True)
            ok = ok and same and np.array_equal(td[p], dt.astype(bool)) and np.array_equal(tc[p], ct.astype(bool)) \
                and np.array_equal(truth[p], (dt * ct).astype(bool))
        cases += 1
        if not ok:
            print(f"FAIL case {cases}: n={n} m={m} variant={variant} shared={shared} crop={None if crop is None else crop.tolist()} "
                  f"seed={seed}", flush=True)
            sys.exit(1)
        if cases % 10 == 0:
            print(f"{cases} cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"all {cases} cases ok")


if __name__ == "__main__":
    main()
