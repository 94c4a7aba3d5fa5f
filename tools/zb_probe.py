"""Timing probe for the tiled z-buffer pass: which outputs cost what (development tool)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea
from pose_estimation_amd import _abi

n, P = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 720
rs = np.random.RandomState(100)
pts_np = rs.uniform(-0.4, 0.4, (n, 3))
if os.environ.get('ZB_MODEL') == 'chair':
    pts_np = pea.run_proj.sample_surface(*pea.run_proj.chair_mesh(), n, seed=100)
pts_np = pts_np[pea.run_proj.spatial_order(pts_np, curve=os.environ.get('ZB_CURVE', 'hilbert'))]
pts = torch.from_numpy(pts_np).cuda()
poses = np.array([[az, el, 0.0, 3.0] for az in range(360) for el in range(-10, 40, 5)], dtype=np.float64)
poses = np.resize(poses, (P, 4))
cams = torch.from_numpy(pea.pt2render.camera_matrices(poses)).cuda()
truth = torch.empty((P, n), dtype=torch.uint8, device="cuda")
packed = torch.empty((P, 540, 960), dtype=torch.int64, device="cuda")
need = _abi.lib.ispa_project_zbuffer_workspace_bytes(n, P, 540, 960)
ws = torch.empty(need, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream

def run(pk, tr, reps=10):
    f = lambda: _abi.check(_abi.lib.ispa_project_zbuffer(pts.data_ptr(), n, cams.data_ptr(), P, 540, 960, 0, 0.01,
                                                         pk, None, None, tr, ws.data_ptr(), need, st))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print("packed+truth %.3f ms" % run(packed.data_ptr(), truth.data_ptr()))
print("packed only  %.3f ms" % run(packed.data_ptr(), None))
print("truth only   %.3f ms" % run(None, truth.data_ptr()))
# visits per tile, from the box table
box = ws[(_abi.lib.ispa_project_zbuffer_workspace_bytes(n, 0, 540, 960) // 256) * 256:]
nch = (n + 63) // 64
off = ((nch * 48 + 255) // 256) * 256
b = ws[off:off + P * nch * 8].view(torch.int16).view(P, nch, 4).cpu().numpy().astype(np.int64)
w = b[..., 1] - b[..., 0] + 1
h = b[..., 3] - b[..., 2] + 1
print("box width mean %.1f max %d, height mean %.1f max %d" % (w.mean(), w.max(), h.mean(), h.max()))
tx = (np.minimum(b[..., 1], 959) // 128 - np.maximum(b[..., 0], 0) // 128 + 1).clip(0)
ty = (np.minimum(b[..., 3], 539) // 64 - np.maximum(b[..., 2], 0) // 64 + 1).clip(0)
print("chunk visits per pose: %.0f (chunks %d)" % ((tx * ty).sum(1).mean(), nch))

import ctypes
if hasattr(_abi.lib, "ispa_dev_zb_stamps"):
    buf = (ctypes.c_ulonglong * 16)()
    _abi.lib.ispa_dev_zb_stamps(buf, 1)
    run(packed.data_ptr(), truth.data_ptr(), reps=1)
    # run() does 3 warm-up + 1 timed call
    _abi.lib.ispa_dev_zb_stamps(buf, 1)
    calls = 4                                    # run() = 3 warm-up calls + 1 timed
    active = max(buf[6], 1)
    idle = max(calls * P * 72 - buf[6], 1)
    print("active tiles per call %.0f of %d, chunk visits per active tile %.1f" % (buf[6] / calls, P * 72, buf[7] / active))
    for k, nm in enumerate(["list", "init", "pass1", "pass2", "writeout"]):
        print("  %-9s %8.0f shader clocks per active tile" % (nm, buf[k] / active))
    print("  idle tiles: list %.0f, writeout %.0f shader clocks" % (buf[8] / idle, buf[9] / idle))
