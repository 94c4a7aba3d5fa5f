import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import pose_estimation_amd as pea
from pose_estimation_amd import disparity, run_proj
verts, faces = run_proj.chair_mesh()
for n in (10000, 100000, 1000000):
    pts = torch.from_numpy(run_proj.sample_surface(verts, faces, n, seed=1)).cuda()
    for m in (1, 8):
        params = np.stack([[a * 40.0, 10.0, 0.0, 3.0] for a in range(m)])
        depth = torch.full((540, 960), 3.0, dtype=torch.float32, device="cuda")
        f = lambda: disparity.demo_flow(pts, params, depth)
        for _ in range(2): f()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(3): f()
        t1.record(); torch.cuda.synchronize()
        print(n, m, f"{t0.elapsed_time(t1)/3:.3f} ms", flush=True)
