"""Development probe: time the geometry / splat kernels at BASELINE config sizes (GPU box)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pose_estimation_amd as pea
from pose_estimation_amd import _abi, _device as dv

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

# correlation maps at the reference's size (B=8, C=128, 56x56 -> 28x28, K=10)
def corr_probe():
    f1 = torch.randn(8, 128, 56, 56, device="cuda"); f1 = f1 / f1.norm(dim=1, keepdim=True)
    f2 = torch.randn(8, 128, 56, 56, device="cuda"); f2 = f2 / f2.norm(dim=1, keepdim=True)
    knn = torch.randint(0, 28, (10, 2), device="cuda")
    mask = (torch.rand(8, 1, 56, 56, device="cuda") > 0.3).float()
    layer = pea.correspondence.CorrelationMapLayer()
    ms = timeit(lambda: layer.forward(f1, f2, knn), 20)
    print(f"corr eccv  B=8 K=10: {ms*1e3:.1f} us  ({8/ms*1e3:.0f} pairs/s)")
    ms = timeit(lambda: pea.correspondence.correlation_map_acmm(f1, f2, mask), 20)
    print(f"corr acmm  B=8     : {ms*1e3:.1f} us  ({8/ms*1e3:.0f} pairs/s, {2*784*784*128*8/ms/1e9:.1f} TFLOP/s)")
    ms = timeit(lambda: pea.correspondence.l2norm(f1), 20)
    print(f"l2norm     B=8     : {ms*1e3:.1f} us  ({2*f1.numel()*4/ms/1e6:.0f} GB/s)")


if len(sys.argv) > 1 and sys.argv[1] == "corr":
    corr_probe()
    sys.exit(0)

N = 100000
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
assert P >= 512, "the splat section slices 2 x 256 poses out of the projected array"
rs = np.random.RandomState(0)
pts = torch.from_numpy(rs.uniform(-0.4, 0.4, (N, 3))).cuda()
poses = np.array([[az, el, 0, 3.0] for az in range(360) for el in range(-10, 40, 5)], dtype=np.float64)[:P]
cams = torch.from_numpy(pea.pt2render.camera_matrices(poses)).cuda()
out = torch.empty((P, 3, N), dtype=torch.float64, device="cuda")
truth = torch.empty((P, N), dtype=torch.uint8, device="cuda")
depth = torch.full((540, 960), 3.0, dtype=torch.float32, device="cuda")
crop = torch.tensor([[100, 400, 300, 700]], dtype=torch.int32, device="cuda")
st = dv.stream_ptr()
ms = timeit(lambda: _abi.check(_abi.lib.ispa_project(pts.data_ptr(), N, cams.data_ptr(), P, out.data_ptr(), st)))
print(f"project           P={P}: {ms:.3f} ms  {(24*N + 24*N*P)/ms/1e6:.1f} GB/s")
for variant in (0, 1):
    ms = timeit(lambda: _abi.check(_abi.lib.ispa_project_prune(pts.data_ptr(), N, cams.data_ptr(), P, depth.data_ptr(), 0, 540, 960, 540, 950, 0.15, variant, crop.data_ptr(), 0, out.data_ptr(), truth.data_ptr(), None, None, st)))
    print(f"project_prune v{variant}  P={P}: {ms:.3f} ms  {(24*N + 29*N*P)/ms/1e6:.1f} GB/s (algorithmic 24N+29NP)")
# z-buffer on a subset of poses
PZ = min(P, 64)
packed = torch.empty((PZ, 540, 960), dtype=torch.int64, device="cuda")
zdepth = torch.empty((PZ, 540, 960), dtype=torch.float32, device="cuda")
winner = torch.empty((PZ, 540, 960), dtype=torch.int32, device="cuda")
ms = timeit(lambda: _abi.check(_abi.lib.ispa_zbuffer(out.data_ptr(), N, PZ, 540, 960, 0, packed.data_ptr(), winner.data_ptr(), zdepth.data_ptr(), st)))
print(f"zbuffer 540x960   P={PZ}: {ms:.3f} ms  {(24*N*PZ + 2*540*960*8*PZ)/ms/1e6:.1f} GB/s (algorithmic)")
ms = timeit(lambda: _abi.check(_abi.lib.ispa_zbuffer_visibility(out.data_ptr(), N, PZ, zdepth.data_ptr(), 540, 960, 0.01, truth.data_ptr(), st)))
print(f"zbuffer_vis       P={PZ}: {ms:.3f} ms  {(29*N*PZ)/ms/1e6:.1f} GB/s")
# flow splat batch
M = 256
uv1 = out[:M].contiguous(); uv2 = out[M:2*M].contiguous() if P >= 2*M else out[:M].contiguous()
tr = (torch.rand((M, N), device="cuda") < 0.45).to(torch.uint8)
_abi.check(_abi.lib.ispa_crop_remap(uv1.data_ptr(), N, M, crop.data_ptr(), 0, 1, None, st))
_abi.check(_abi.lib.ispa_crop_remap(uv2.data_ptr(), N, M, crop.data_ptr(), 0, 1, None, st))
owner = torch.empty((M, 224*224), dtype=torch.int32, device="cuda")
flow = torch.empty((M, 224, 224, 2), dtype=torch.float64, device="cuda")
ms = timeit(lambda: _abi.check(_abi.lib.ispa_flow_splat(uv1.data_ptr(), uv2.data_ptr(), tr.data_ptr(), N, M, 2, 224, owner.data_ptr(), flow.data_ptr(), st)))
print(f"flow_splat        M={M}: {ms:.3f} ms  {M/ms*1e3:.0f} maps/s  {(33*N + 802816)*M/ms/1e6:.1f} GB/s (algorithmic 33N+802816)")
