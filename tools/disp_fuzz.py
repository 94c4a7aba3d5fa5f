"""One-off randomized differential check of make_disparity_maps (fused and un-fused) against the CPU oracle
(development tool, needs a GPU): random models, poses, crops (tall, wide, partly outside the image, tiny)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea                       # noqa: E402
from oracle import disparity as odisp                   # noqa: E402
from oracle import geometry as og                       # noqa: E402
from oracle import zbuffer as ozb                       # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    rs = np.random.RandomState(seed)
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        n = int(rs.randint(1, 20000))
        if rs.rand() < 0.5:
            pts = pea.run_proj.sample_surface(*pea.run_proj.chair_mesh(), n, seed=int(rs.randint(1 << 30)))
        else:
            pts = rs.uniform(-0.4, 0.4, (n, 3))
        m = int(rs.randint(1, 4))
        params = np.stack([rs.uniform(0, 360, m), rs.uniform(-10, 40, m), rs.uniform(-10, 10, m),
                           rs.choice([1.5, 2.6, 3.0, 4.0], m)], axis=1)
        depths, crops = [], []
        for p in params:
            r = og.get_3d_to_2d(pts, p)
            with np.errstate(invalid="ignore", over="ignore"):
                d = ozb.zbuffer(r, 540, 960, radius=2)[2]
            d[~np.isfinite(d)] = 20.0
            depths.append(d.astype(np.float32))
            style = int(rs.randint(0, 4))
            if style == 0:                                # around the object
                t = og.depth_visibility_array_a(r, depths[-1])[0]
                if t.any():
                    c = [int(r[1, t].min()) - 3, int(r[1, t].max()) + 3, int(r[0, t].min()) - 3, int(r[0, t].max()) + 3]
                else:
                    c = [100, 400, 300, 700]
            elif style == 1:                              # random box, may leave the image
                y0, x0 = int(rs.randint(-50, 400)), int(rs.randint(-50, 800))
                c = [y0, y0 + int(rs.randint(2, 500)), x0, x0 + int(rs.randint(2, 700))]
            elif style == 2:                              # tiny
                y0, x0 = int(rs.randint(200, 300)), int(rs.randint(400, 500))
                c = [y0, y0 + int(rs.randint(2, 12)), x0, x0 + int(rs.randint(2, 12))]
            else:                                         # whole image
                c = [0, 540, 0, 960]
            crops.append(c)
        got = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops))
        unfused = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops), fused=False)
        ok = np.array_equal(got, unfused)
        # the same maps with the points processed in another order (curve order, or a random one), ranked as given
        perm = None if rs.rand() < 0.5 else rs.permutation(len(pts))
        ordered = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops),
                                                    point_order=pea.disparity.PointOrder(pts, perm))
        ok = ok and np.array_equal(got, ordered)
        for k in range(m):
            want = odisp.make_disparity_core(pts, params[k].tolist(), depths[k], crops[k])
            ok = ok and np.array_equal(got[k], want)
        cases += 1
        if not ok:
            print(f"FAIL case {cases}: n={n} m={m} crops={crops} seed={seed}", flush=True)
            sys.exit(1)
        if cases % 10 == 0:
            print(f"{cases} cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"all {cases} cases ok")


if __name__ == "__main__":
    main()
