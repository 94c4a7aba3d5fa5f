"""One-off randomized differential check of the fused z-buffer pass against the CPU oracle (development tool,
needs a GPU): random image sizes (odd widths, partial tiles), pose counts (scheduling tail), splat radii, models
with duplicate points (equal depths: the lowest index wins) and points that cannot take part."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea                       # noqa: E402
from oracle import cnative                              # noqa: E402
from oracle import zbuffer as ozb                       # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    rs = np.random.RandomState(seed)
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        n = int(rs.randint(1, 6000))
        h, w = int(rs.randint(8, 300)), int(rs.randint(8, 400))
        if rs.rand() < 0.3:
            h, w = 540, 960
        radius = int(rs.choice([0, 0, 1, 2]))
        n_pose = int(rs.randint(1, 12))
        kind = int(rs.randint(0, 3))
        if kind == 0:
            pts = rs.uniform(-0.45, 0.45, (n, 3))
        elif kind == 1:
            pts = pea.run_proj.sample_surface(*pea.run_proj.chair_mesh(), n, seed=int(rs.randint(1 << 30)))
        else:                                             # few distinct points, many duplicates
            pts = rs.uniform(-0.4, 0.4, (max(1, n // 50), 3))[rs.randint(0, max(1, n // 50), n)]
        if n > 10:
            pts[rs.randint(0, n)] = [np.nan, 0.0, 0.0]
            pts[rs.randint(0, n)] *= 50.0
        if rs.rand() < 0.5:
            pts = pts[pea.run_proj.spatial_order(np.nan_to_num(pts))]
        poses = np.stack([rs.uniform(0, 360, n_pose), rs.uniform(-20, 50, n_pose), rs.uniform(-15, 15, n_pose),
                          rs.choice([0.5, 1.5, 3.0, 6.0], n_pose)], axis=1)
        cams = pea.pt2render.camera_matrices(poses)
        cams[:, 0] += cams[:, 2] * (w / 2 - 480)          # principal point into the image
        cams[:, 1] += cams[:, 2] * (h / 2 - 270)
        relax = float(rs.choice([0.0, 0.01, 0.15]))
        got = pea.zbuffer.project_zbuffer(pts, cams, (h, w), radius=radius, relax=relax,
                                          want=("packed", "winner", "depth", "truth"))
        for p in range(n_pose):
            render = cnative.project_fma(pts, cams[p])
            with np.errstate(invalid="ignore", over="ignore"):
                key, win, depth = ozb.zbuffer(render, h, w, radius=radius)
                vis = ozb.zbuffer_visibility(render, depth, relax=relax)
            ok = (np.array_equal(got["packed"][p].view(np.uint64), key) and np.array_equal(got["winner"][p], win)
                  and np.array_equal(got["depth"][p], depth) and np.array_equal(got["truth"][p], vis))
            if not ok:
                print(f"FAIL case {cases}: n={n} size=({h},{w}) radius={radius} poses={n_pose} kind={kind} pose {p} "
                      f"seed={seed}", flush=True)
                sys.exit(1)
        cases += 1
        if cases % 20 == 0:
            print(f"{cases} cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"all {cases} cases ok")


if __name__ == "__main__":
    main()
