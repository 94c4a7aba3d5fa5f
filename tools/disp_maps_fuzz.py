"""One-off randomized check of make_disparity's one-workgroup-per-map kernel (development tool, needs a GPU): a
call with as many maps as the chip has compute units against the same maps in two calls of half the size (which
take the kernels with per-band lists, themselves checked against the CPU oracle by tools/disp_fuzz.py), for
random point counts, spreads of depth, crops and point orders.  python tools/disp_maps_fuzz.py [seed] [seconds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea                       # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    rs = np.random.RandomState(seed)
    maps = torch.cuda.get_device_properties(0).multi_processor_count
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        n = int(rs.choice([1, 63, 64, 1000, 4095, 4096, 4097, int(rs.randint(1, 30000))]))
        if rs.rand() < 0.5 and n > 10:
            pts = pea.run_proj.sample_surface(*pea.run_proj.chair_mesh(), n, seed=int(rs.randint(1 << 30)))
        else:
            pts = rs.uniform(-0.4, 0.4, (n, 3))
        m = int(rs.choice([maps, maps - int(rs.randint(0, maps // 8)), 2 * maps]))
        params = np.stack([rs.uniform(0, 360, m), rs.uniform(-10, 40, m), rs.uniform(-10, 10, m),
                           rs.choice([1.5, 2.6, 3.0, 4.0], m)], axis=1)
        coarse = torch.from_numpy(rs.uniform(-0.3, 0.3, (m, 54, 96)).astype(np.float32)).cuda()
        depth = torch.from_numpy(params[:, 3].astype(np.float32)).cuda()[:, None, None] + \
            torch.repeat_interleave(torch.repeat_interleave(coarse, 10, dim=1), 10, dim=2)
        style = int(rs.randint(0, 3))
        if style == 0:
            crops = np.stack([rs.randint(60, 180, m), rs.randint(330, 480, m), rs.randint(250, 400, m),
                              rs.randint(560, 720, m)], axis=1)
        elif style == 1:                                  # partly outside the image, tall or wide
            crops = np.stack([rs.randint(-50, 250, m), rs.randint(260, 600, m), rs.randint(-50, 450, m),
                              rs.randint(460, 1000, m)], axis=1)
        else:                                             # tiny
            y0, x0 = rs.randint(200, 300, m), rs.randint(400, 500, m)
            crops = np.stack([y0, y0 + rs.randint(2, 30, m), x0, x0 + rs.randint(2, 30, m)], axis=1)
        order = pea.disparity.PointOrder(pts) if rs.rand() < 0.5 else None
        pts_t = torch.from_numpy(pts).cuda()
        whole = pea.disparity.make_disparity_maps(pts_t, params, depth, crops, point_order=order)
        cut = [0, m // 3, 2 * m // 3, m]                  # thirds: below 85 % of the compute units each
        parts = [pea.disparity.make_disparity_maps(pts_t, params[a:b], depth[a:b], crops[a:b], point_order=order)
                 for a, b in zip(cut[:-1], cut[1:])]
        if not torch.equal(whole, torch.cat(parts)):
            print("MISMATCH seed", seed, "case", cases, "n", n, "m", m, "style", style, "ordered", order is not None)
            sys.exit(1)
        cases += 1
        if cases % 10 == 0:
            print(cases, "cases ok (%.0f s)" % (time.time() - t0), flush=True)
    print("all", cases, "cases ok")


if __name__ == "__main__":
    main()
