"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Bars: integer / boolean / index results bit-exact; float64 z and flow values bit-exact against
the oracle's fixed-order twin and <= 1e-12 relative against the imported reference's numpy;
float32 distances and correlation maps within 1e-4 relative (tolerances written at each assert).
"""
import os

import numpy as np
import pytest

from oracle import cnative, correlation as ocorr, disparity as odisp, geometry as og, keypoints as okp
from oracle import nn as onn, zbuffer as ozb

pytestmark = pytest.mark.gpu

ANCHORS = ["bed", "chair", "sofa_b", "sofa", "table"]


@pytest.fixture(scope="module")
def pea():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import pose_estimation_amd as p
    return p


# ---------------------------------------------------------------- geometry
def test_camera_host_functions(pea, golden_geometry):
    z = golden_geometry
    for pose, e, p, r in zip(z["g1_poses"], z["g1_E"], z["g1_P"], z["g1_R"]):
        assert np.array_equal(pea.pt2render.rot_three_angles(pose[0], pose[1], pose[2]), r)
        assert np.array_equal(pea.pt2render.ex_camera_matrix(*pose), e)
        assert np.array_equal(pea.run_proj.camera_from_params(pose), p)
    batch = pea.pt2render.camera_matrices(z["g1_poses"])
    assert np.allclose(batch, z["g1_P"], rtol=1e-13, atol=1e-12)


def test_projection_golden(pea, golden_geometry):
    z = golden_geometry
    cam = z["g2_big_cam"]
    for name in ANCHORS:
        got = pea.pt2render.get_render_location(z[f"g2_anchor_{name}"], cam)
        assert np.array_equal(got, z[f"g2_anchor_{name}_render"])
    got = pea.pt2render.get_render_location(z["g2_obj_vertices"], cam)
    assert np.array_equal(got[:2], z["g2_obj_render"][:2])
    assert np.allclose(got[2], z["g2_obj_render"][2], rtol=1e-12, atol=0)
    assert np.array_equal(got, cnative.project_fma(z["g2_obj_vertices"], cam))     # bit-exact twin


def test_projection_100k_bit_exact(pea, golden_geometry):
    z = golden_geometry
    cam = z["g2_big_cam"]
    big = np.random.RandomState(0).uniform(-0.4, 0.4, (100000, 3))
    got = pea.pt2render.get_render_location(big, cam)
    assert np.array_equal(got, cnative.project_fma(big, cam))
    sub = z["g2_big_sub_index"]
    assert np.array_equal(got[:2, sub], z["g2_big_sub_render"][:2])
    assert np.allclose(got[2, sub], z["g2_big_sub_render"][2], rtol=1e-12, atol=0)
    import hashlib
    digest = hashlib.sha256(np.ascontiguousarray(got[:2]).tobytes()).digest()
    assert np.array_equal(np.frombuffer(digest, dtype=np.uint8), z["g2_big_uv_sha256"])


def test_projection_half_integer_quotients(pea):
    """x / z landing exactly on (or one ulp beside) k + 0.5 must round like numpy: half to even."""
    cam = np.array([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 0, 2.0]])
    k = np.arange(-2000, 2000, dtype=np.float64)
    x = 2 * k + 1                                  # x / 2 = k + 0.5 exactly
    pts = np.stack([x, np.nextafter(x, np.inf), np.nextafter(x, -np.inf)], axis=1)
    pts = np.concatenate([pts, pts[:, ::-1] * 3.0, np.array([[1e300, -1e300, 0.0], [np.inf, 1.0, 2.0]])])
    cams = np.stack([cam, cam * np.array([[3.0], [0.5], [1.0]])])
    cams[1, 2, 3] = 6.0                            # z = 6: quotients k/2 + 1/4 steps, some at .5
    got = pea.pt2render.get_render_location(pts, cams)
    for p in range(2):
        want = cnative.project_fma(pts, cams[p])
        assert np.array_equal(got[p], want, equal_nan=True)
        with np.errstate(invalid="ignore", over="ignore"):
            assert np.array_equal(got[p][:2], og.get_render_location(pts, cams[p])[:2], equal_nan=True)


def test_projection_batched_and_ragged(pea, golden_geometry):
    z = golden_geometry
    cams = z["g1_P"][:37]
    for n in (0, 1, 255, 257, 1000):
        pts = np.random.RandomState(n).uniform(-0.5, 0.5, (n, 3))
        got = pea.pt2render.get_render_location(pts, cams)
        assert got.shape == (37, 3, n)
        for p in (0, 17, 36):
            assert np.array_equal(got[p], cnative.project_fma(pts, cams[p]))


@pytest.mark.parametrize("tag", ["near", "std", "tilt"])
@pytest.mark.parametrize("dname", ["surface", "const", "plane"])
def test_depth_visibility_golden(pea, golden_geometry, tag, dname):
    z = golden_geometry
    r = z[f"g3_{tag}_render"]
    d = np.full((540, 960), 3.0, np.float32) if dname == "const" else z[f"g3_{tag}_depth_{dname}"]
    arr = r.copy()
    ta, same = pea.pt2render.depth_visibility_array(arr, d)
    assert same is arr and np.array_equal(arr, r)
    assert ta.dtype == np.bool_ and np.array_equal(ta, z[f"g3_{tag}_{dname}_truth_a"])
    tb, pruned = pea.pt2render_b.depth_visibility_array(r.copy(), d)
    assert np.array_equal(tb, z[f"g3_{tag}_{dname}_truth_b"])
    assert np.array_equal(pruned, z[f"g3_{tag}_{dname}_pruned_b"])


def test_depth_visibility_edges(pea, golden_geometry):
    z = golden_geometry
    r, d = z["g3_edge_render"], z["g3_edge_depth"]
    assert np.array_equal(pea.pt2render.depth_visibility_array(r.copy(), d)[0], z["g3_edge_truth_a"])
    assert np.array_equal(pea.pt2render_b.depth_visibility_array(r.copy(), d)[0], z["g3_edge_truth_b"])
    bad = np.array([[np.nan, np.inf, 1e300, 100.0], [10.0, 10.0, 10.0, np.nan], [2.5] * 4])
    assert not pea.pt2render.depth_visibility_array(bad, d)[0].any()


@pytest.mark.parametrize("tag", ["wide", "tall", "square", "odd"])
def test_crop_remap_golden(pea, golden_geometry, tag):
    z = golden_geometry
    arr = z["g3_std_render"].copy()
    t, same = pea.pt2render.crop_visibility_array(arr, z[f"g4_{tag}_crop"].tolist())
    assert same is arr                                     # the caller's array is rewritten
    assert np.array_equal(t, z[f"g4_{tag}_truth"])
    assert np.array_equal(arr, z[f"g4_{tag}_render"])


def test_crop_noresize(pea, golden_geometry):
    z = golden_geometry
    arr = z["g3_std_render"].copy()
    t, _ = pea.pt2render.crop_visibility_array(arr, [100, 400, 300, 700], resize=False)
    assert np.array_equal(t, z["g4_noresize_truth"]) and np.array_equal(arr, z["g4_noresize_render"])


def test_prune_pts_and_fused(pea, golden_geometry):
    z = golden_geometry
    pts, params = z["g3_std_pts"], z["g3_std_params"].tolist()
    depth = z["g3_std_depth_surface"]
    crop = [100, 400, 300, 700]
    want_r = og.get_3d_to_2d(pts, params)
    want_t, want_c = og.prune_pts(want_r, depth, crop=True, crop_params=crop)
    got_r = pea.run_proj.get_3d_to_2d(pts, params)
    got_t, got_c = pea.run_proj.prune_pts(got_r, depth, crop=True, crop_params=crop)
    assert got_c is got_r and np.array_equal(got_t, want_t) and np.array_equal(got_c[:2], want_c[:2])
    # no crop: float truth like the reference (ones * bool)
    t2, _ = pea.run_proj.prune_pts(pea.run_proj.get_3d_to_2d(pts, params), depth)
    assert np.array_equal(t2, og.prune_pts(og.get_3d_to_2d(pts, params), depth)[0])
    # fused kernel, several poses, per-pose depth and crop
    poses = [params, [200.0, -5.0, 12.0, 2.0], [30.0, 10.0, 0.0, 1.1]]
    cams = np.stack([og.camera_from_params(p) for p in poses])
    depths = np.stack([z["g3_std_depth_surface"], z["g3_tilt_depth_surface"], z["g3_near_depth_surface"]])
    crops = np.array([crop, [50, 500, 400, 600], [101, 388, 297, 641]])
    for variant in ("A", "B"):
        t, c, td, tc = pea.run_proj.project_prune(pts, cams, depths, crops, variant=variant, want_parts=True)
        for p in range(3):
            r = cnative.project_fma(pts, cams[p])
            vis = (og.depth_visibility_array_a if variant == "A" else og.depth_visibility_array_b)(r, depths[p])[0]
            ct, rc = og.crop_visibility_array(r, crops[p].tolist())
            assert np.array_equal(td[p], vis) and np.array_equal(tc[p], ct)
            assert np.array_equal(t[p], vis & ct) and np.array_equal(c[p], rc)


@pytest.mark.parametrize("name", ANCHORS)
def test_get_render_array_variant_b(pea, golden_disparity, golden_geometry, name):
    z = golden_disparity
    fd = {"azimuth": 120.0, "elevation": 10.0, "tilt": 0.0, "rho": 2.2, "crop_params": [60, 480, 250, 720]}
    truth, crop_pts = pea.io_utils.get_render_array(fd, golden_geometry[f"g2_anchor_{name}"], z[f"rb_{name}_depth"])
    assert np.array_equal(truth, z[f"rb_{name}_truth"])
    assert np.array_equal(crop_pts[:2], z[f"rb_{name}_crop_pts"][:2])
    assert np.allclose(crop_pts[2], z[f"rb_{name}_crop_pts"][2], rtol=1e-12, atol=0)


def test_get_pt_and_grid_indices(pea, golden_disparity, golden_geometry):
    name = "03001627_13fdf00cde077f562f6f52615fb75fca_a120@000_e10@000_t0@000_d2@200_60_480_250_720_.jpg"
    anchors = golden_geometry["g2_anchor_chair"]
    depth = golden_disparity["rb_chair_depth"]
    got = pea.correspondence.get_pt(anchors, name, depth)
    want = okp.get_pt(anchors, name, depth)
    assert got.shape == (1, 10, 3) and np.array_equal(got, want)
    assert np.array_equal(pea.correspondence.knn_inds_to_grid(got[0, :, :2]), okp.knn_inds_to_grid(want[0, :, :2]))


def test_spatial_order_is_a_pure_permutation(pea):
    rs = np.random.RandomState(3)
    pts = rs.uniform(-0.4, 0.4, (20000, 3))
    perm = pea.run_proj.spatial_order(pts)
    assert sorted(perm.tolist()) == list(range(20000))
    cams = pea.pt2render.camera_matrices(np.array([[30.0, 10.0, 0.0, 3.0], [200.0, -5.0, 3.0, 2.5]]))
    depth = np.full((540, 960), 3.0, np.float32)
    t0, r0 = pea.run_proj.project_prune(pts, cams, depth, [100, 400, 300, 700])
    t1, r1 = pea.run_proj.project_prune(pts[perm], cams, depth, [100, 400, 300, 700])
    assert np.array_equal(t1, t0[:, perm]) and np.array_equal(r1, r0[:, :, perm])
    # neighbours in the array are neighbours on screen: mean pixel step shrinks by a large factor
    step = lambda r: np.abs(np.diff(r[0, 0])).mean()
    assert step(r1) < 0.2 * step(r0)
    assert len(pea.run_proj.spatial_order(np.zeros((0, 3)))) == 0


# ---------------------------------------------------------------- splat writers
@pytest.mark.parametrize("tag", ["cloud", "mesh", "edge"])
def test_make_disparity_golden(pea, golden_disparity, tag):
    z = golden_disparity
    for fused in (True, False):
        flow = pea.disparity.make_disparity_maps(z[f"{tag}_pts"], z[f"{tag}_params"], z[f"{tag}_depth"],
                                                 z[f"{tag}_crop"], fused=fused)
        assert flow.shape == (1, 224, 224, 2) and flow.dtype == np.float64
        assert np.array_equal(flow[0], z[f"{tag}_flow"])


@pytest.mark.parametrize("ordered,spread", [(False, 2), (True, 2), (True, 1), (False, 3), (True, 4), (False, 5), (True, 0)])
def test_make_disparity_one_workgroup_per_map(pea, ordered, spread):
    """A call with as many maps as the chip has compute units takes the kernel that gives each map a workgroup of
    its own (spreads 1 to 4); the same maps in two calls of half the size take the kernels with per-band lists.
    Same bytes, the bytes of the path that materialises the projected coordinates, and (at the reference's spread)
    the oracle's flow for a few of the maps."""
    import torch
    maps = torch.cuda.get_device_properties(0).multi_processor_count
    rs = np.random.RandomState(5)
    n = 9000                                                   # three rounds of 4096 points, the last one ragged
    pts = np.concatenate([pea.run_proj.sample_surface(*pea.run_proj.chair_mesh(), n - 2000, seed=2),
                          rs.uniform(-0.4, 0.4, (2000, 3))])
    params = np.stack([rs.uniform(0, 360, maps), rs.uniform(-10, 40, maps), rs.uniform(-5, 5, maps),
                       rs.choice([2.6, 3.0], maps)], axis=1)
    # depth "images": the camera distance plus a coarse random offset per texel block, so that visibility varies
    coarse = torch.from_numpy(rs.uniform(-0.25, 0.25, (maps, 54, 96)).astype(np.float32)).cuda()
    depth = torch.from_numpy(params[:, 3].astype(np.float32)).cuda()[:, None, None] + \
        torch.repeat_interleave(torch.repeat_interleave(coarse, 10, dim=1), 10, dim=2)
    crops = np.stack([rs.randint(60, 180, maps), rs.randint(330, 480, maps), rs.randint(250, 400, maps),
                      rs.randint(560, 720, maps)], axis=1)
    order = pea.disparity.PointOrder(pts) if ordered else None
    pts_t = torch.from_numpy(pts).cuda()
    whole = pea.disparity.make_disparity_maps(pts_t, params, depth, crops, point_order=order, spread=spread)
    half = maps // 2
    parts = [pea.disparity.make_disparity_maps(pts_t, params[a:b], depth[a:b], crops[a:b], point_order=order,
                                               spread=spread) for a, b in ((0, half), (half, maps))]
    assert torch.equal(whole, torch.cat(parts))
    assert torch.equal(whole, pea.disparity.make_disparity_maps(pts_t, params, depth, crops, fused=False, spread=spread))
    whole = whole.cpu().numpy()
    depth_np = depth[:3].cpu().numpy()
    for m in range(3 if spread == 2 else 0):
        want = odisp.make_disparity_core(pts, params[m].tolist(), depth_np[m], crops[m])
        assert np.array_equal(whole[m], want)
        assert np.count_nonzero(want) > 2000


def test_make_disparity_one_workgroup_per_map_bad_ranks_and_workspace(pea):
    """The kernel with a workgroup per map through the C ABI: ranks and inverse entries outside [0, n) give the
    maps of the kernels with lists (points dropped, pixels left empty, nothing faults); a workspace that is too
    small or not 8-byte aligned is refused; a single point and no point at all work."""
    import torch
    from pose_estimation_amd import _abi, _device as dv
    maps = torch.cuda.get_device_properties(0).multi_processor_count
    rs = np.random.RandomState(11)
    n = 5000
    pts = pea.run_proj.sample_surface(*pea.run_proj.chair_mesh(), n, seed=4)
    params = np.stack([rs.uniform(0, 360, maps), rs.uniform(-10, 40, maps), np.zeros(maps), np.full(maps, 3.0)], axis=1)
    more = params.copy()
    more[:, :2] += 10
    cams1 = dv.to_dev(pea.pt2render.camera_matrices(params), torch.float64)
    cams2 = dv.to_dev(pea.pt2render.camera_matrices(more), torch.float64)
    depth = torch.full((1, 540, 960), 3.0, dtype=torch.float32, device="cuda")
    crops = dv.to_dev(np.tile(np.array([[100, 400, 300, 700]]), (maps, 1)), torch.int32)
    order = pea.disparity.PointOrder(pts)
    rank = order.rank.clone()
    rank[::7] = n + 5
    rank[3::7] = -2
    inv = order.inverse.clone()
    inv[::5] = n
    inv[2::5] = -1

    def call(first, count, ws, need, points=order.points, n_=n, rk=rank, iv=inv):
        flow = torch.full((count, 224, 224, 2), np.nan, dtype=torch.float64, device="cuda")
        rc = _abi.lib.ispa_make_disparity_ordered(
            dv.ptr(points), dv.ptr(rk) if rk is not None else None, dv.ptr(iv) if iv is not None else None, n_,
            dv.ptr(cams1[first:]), dv.ptr(cams2[first:]), count, dv.ptr(depth), 0, 540, 960, 540, 950, 0.15,
            dv.ptr(crops[first:]), 1, 2, dv.ptr(flow), dv.ptr(ws) if ws is not None else None, need, dv.stream_ptr())
        torch.cuda.synchronize()
        return rc, flow

    need = _abi.lib.ispa_make_disparity_workspace_bytes(n, maps)
    ws = dv.empty((need + 8,), torch.uint8)
    rc, whole = call(0, maps, ws, need)
    assert rc == 0 and torch.isfinite(whole).all() and torch.count_nonzero(whole) > 1000 * maps
    half = maps // 2
    parts = [call(a, b - a, ws, need)[1] for a, b in ((0, half), (half, maps))]
    assert torch.equal(whole, torch.cat(parts))
    assert call(0, maps, ws, need - 1)[0] == -3                       # ISPA_EWORKSPACE
    assert call(0, maps, ws[4:], need)[0] == -1                       # not 8-byte aligned
    # one point; no point (every map empty)
    one = dv.to_dev(pts[:1], torch.float64)
    rc, flow1 = call(0, maps, ws, need, points=one, n_=1, rk=None, iv=None)
    assert rc == 0 and torch.isfinite(flow1).all()
    rc, flow0 = call(0, maps, ws, need, points=one, n_=0, rk=None, iv=None)
    assert rc == 0 and torch.count_nonzero(flow0) == 0


def test_make_disparity_batch_100k(pea):
    rs = np.random.RandomState(0)
    pts = rs.uniform(-0.4, 0.4, (100000, 3))
    params = np.array([[30.0, 10.0, 0.0, 3.0], [250.0, 25.0, 5.0, 2.6]])
    depths, crops = [], []
    for p in params:
        r = og.get_3d_to_2d(pts, p)
        d = ozb.zbuffer(r, 540, 960, radius=2)[2]
        d[~np.isfinite(d)] = 20.0
        depths.append(d.astype(np.float32))
        t = og.depth_visibility_array_a(r, depths[-1])[0]
        crops.append([int(r[1, t].min()) - 3, int(r[1, t].max()) + 3, int(r[0, t].min()) - 3, int(r[0, t].max()) + 3])
    got = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops))
    unfused = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops), fused=False)
    assert np.array_equal(got, unfused)
    for m in range(2):
        want = odisp.make_disparity_core(pts, params[m].tolist(), depths[m], crops[m])
        assert np.array_equal(got[m], want)
        assert np.count_nonzero(want) > 20000
    # determinism of the atomic owner pass: two more runs give the same bytes
    again = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops))
    assert np.array_equal(got, again)


@pytest.mark.parametrize("tag", ["cloud", "mesh", "edge"])
def test_make_disparity_processing_order_keeps_the_reference_order(pea, golden_disparity, tag):
    """Points processed along a space-filling curve (or in any other order), ranks = the reference's order: the
    flow maps are those of the reference's order, bit for bit."""
    z = golden_disparity
    pts = z[f"{tag}_pts"]
    args = (z[f"{tag}_params"], z[f"{tag}_depth"], z[f"{tag}_crop"])
    for perm in (None, np.random.RandomState(5).permutation(len(pts)), np.arange(len(pts))[::-1]):
        order = pea.disparity.PointOrder(pts, perm)
        flow = pea.disparity.make_disparity_maps(pts, *args, point_order=order)
        assert np.array_equal(flow[0], z[f"{tag}_flow"])
    with pytest.raises(ValueError):
        pea.disparity.PointOrder(pts, np.zeros(len(pts), dtype=np.int64))
    with pytest.raises(ValueError):
        pea.disparity.make_disparity_maps(pts[:-1], *args, point_order=order)


def test_make_disparity_ordered_100k_and_bad_ranks(pea):
    import torch
    from pose_estimation_amd import _abi, _device as dv
    rs = np.random.RandomState(0)
    pts = rs.uniform(-0.4, 0.4, (100000, 3))
    params = np.array([[30.0, 10.0, 0.0, 3.0], [250.0, 25.0, 5.0, 2.6], [100.0, 0.0, -5.0, 2.2]])
    depths, crops = [], []
    for p in params:
        r = og.get_3d_to_2d(pts, p)
        d = ozb.zbuffer(r, 540, 960, radius=2)[2]
        d[~np.isfinite(d)] = 20.0
        depths.append(d.astype(np.float32))
        t = og.depth_visibility_array_a(r, depths[-1])[0]
        crops.append([int(r[1, t].min()) - 3, int(r[1, t].max()) + 3, int(r[0, t].min()) - 3, int(r[0, t].max()) + 3])
    want = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops))
    order = pea.disparity.PointOrder(pts)
    got = pea.disparity.make_disparity_maps(pts, params, np.stack(depths), np.array(crops), point_order=order)
    assert np.array_equal(got, want) and np.count_nonzero(want) > 60000
    # through the C ABI with ranks outside [0, n) and inverse entries outside it: points dropped, pixels left
    # empty, nothing faults
    n = len(pts)
    bad_rank = order.rank.clone()
    bad_rank[::7] = n + 5
    bad_rank[3::7] = -2
    bad_inv = order.inverse.clone()
    bad_inv[::5] = n
    cams1 = dv.to_dev(np.stack([pea.run_proj.camera_from_params(p) for p in params]), torch.float64)
    more = params.copy()
    more[:, :2] += 10
    cams2 = dv.to_dev(np.stack([pea.run_proj.camera_from_params(p) for p in more]), torch.float64)
    depth_t = dv.to_dev(np.stack(depths), torch.float32)
    crop_t = dv.to_dev(np.array(crops), torch.int32)
    flow = dv.empty((3, 224, 224, 2), torch.float64)
    need = _abi.lib.ispa_make_disparity_workspace_bytes(n, 3)
    ws = dv.empty((need,), torch.uint8)
    _abi.check(_abi.lib.ispa_make_disparity_ordered(
        dv.ptr(order.points), dv.ptr(bad_rank), dv.ptr(bad_inv), n, dv.ptr(cams1), dv.ptr(cams2), 3, dv.ptr(depth_t), 1,
        540, 960, 540, 950, 0.15, dv.ptr(crop_t), 1, 2, dv.ptr(flow), dv.ptr(ws), need, dv.stream_ptr()))
    torch.cuda.synchronize()
    assert np.isfinite(flow.cpu().numpy()).all()
    rc = _abi.lib.ispa_make_disparity_ordered(
        dv.ptr(order.points), dv.ptr(bad_rank), None, n, dv.ptr(cams1), dv.ptr(cams2), 3, dv.ptr(depth_t), 1,
        540, 960, 540, 950, 0.15, dv.ptr(crop_t), 1, 2, dv.ptr(flow), dv.ptr(ws), need, dv.stream_ptr())
    assert rc == -1


def test_crop_tables_may_be_device_tensors(pea):
    """Crop boxes as a list, a numpy array or a tensor already on the GPU give the same results."""
    import torch
    rs = np.random.RandomState(12)
    pts = rs.uniform(-0.35, 0.35, (5000, 3))
    params = np.array([[30.0, 10.0, 0.0, 3.0], [200.0, -5.0, 3.0, 2.5]])
    cams = np.stack([pea.run_proj.camera_from_params(p) for p in params])
    depth = np.full((540, 960), 3.0, dtype=np.float32)
    crops = [[60, 470, 250, 720], [40, 500, 200, 760]]
    t0, p0 = pea.run_proj.project_prune(pts, cams, depth, crops)
    t1, p1 = pea.run_proj.project_prune(pts, cams, depth, torch.tensor(crops, dtype=torch.int64, device="cuda"))
    assert np.array_equal(t0, t1) and np.array_equal(p0, p1)
    f0 = pea.disparity.make_disparity_maps(pts, params, depth, np.asarray(crops))
    f1 = pea.disparity.make_disparity_maps(pts, params, depth, torch.tensor(crops, device="cuda"))
    assert np.array_equal(f0, f1)


def test_script_entry_points_on_a_data_tree(pea, tmp_path, golden_geometry):
    """make_disparity / make_dense_keypoints over a directory laid out like the reference's ../data."""
    gv = pea.global_variables
    saved = {k: getattr(gv, k) for k in ("anchor_loc", "depth_loc", "syn_disparity_loc", "syn_dense_keypoint_loc",
                                         "g_save_file_location_final", "sample_points_loc", "g_shapenet_root_folder")}
    try:
        for k in saved:
            setattr(gv, k, str(tmp_path / k))
            (tmp_path / k).mkdir()
        model, synset = gv.syn_template["chair"]
        rs = np.random.RandomState(3)
        pts = rs.uniform(-0.35, 0.35, (20000, 3))
        np.save(tmp_path / "sample_points_loc" / f"{synset}_{model}.npy", pts)
        anchors = golden_geometry["g2_anchor_chair"]
        np.savetxt(tmp_path / "anchor_loc" / f"{model}.txt", anchors, delimiter=" ")
        renders = []
        for az, el, tilt in ((30.0, 10.0, 0.0), (211.5, 25.25, -4.0), (95.0, -5.0, 3.5)):
            d = {"model_type": synset, "model_md5": model, "azimuth": az, "elevation": el, "tilt": tilt, "rho": 2.5,
                 "crop_params": [60, 470, 250, 720]}
            name = os.path.basename(pea.io_utils.get_final_file_name(d))
            r = og.get_3d_to_2d(np.concatenate([pts, anchors]), [az, el, tilt, 2.5])
            depth = ozb.zbuffer(r, 540, 960, radius=2)[2]
            depth[~np.isfinite(depth)] = 20.0
            depth = depth.astype(np.float32)
            (tmp_path / "g_save_file_location_final" / name).write_bytes(b"")
            np.save(tmp_path / "depth_loc" / (pea.io_utils.img_to_depth_file(name) + ".npy"), depth)
            renders.append((name, d, depth))
        (tmp_path / "g_save_file_location_final" / "04256520_other_a0@000_e0@000_t0@000_d2@000_1_2_3_4_.jpg").write_bytes(b"")
        assert pea.make_disparity_syn.make_disparity("chair") == 3
        one_batch = {name: np.load(tmp_path / "syn_disparity_loc" / (name.split(".")[0] + ".npy")) for name, _, _ in renders}
        # again in batches of two: the next batch's depth files are read while the first is on the GPU
        assert pea.make_disparity_syn.make_disparity("chair", batch=2) == 3
        assert pea.make_dense_pointset_syn.make_dense_keypoints("chair") == 3
        for name, d, depth in renders:
            params = [d["azimuth"], d["elevation"], d["tilt"], d["rho"]]
            flow = np.load(tmp_path / "syn_disparity_loc" / (name.split(".")[0] + ".npy"))
            assert np.array_equal(flow, one_batch[name])
            assert flow.dtype == np.float64 and flow.shape == (224, 224, 2)
            assert np.array_equal(flow, odisp.make_disparity_core(pts, params, depth, d["crop_params"]))
            keys = np.load(tmp_path / "syn_dense_keypoint_loc" / (name.split(".")[0] + ".npy"))
            assert np.array_equal(keys, okp.make_dense_keypoints_core(anchors, "chair", params, depth, d["crop_params"]))
    finally:
        for k, v in saved.items():
            setattr(gv, k, v)


def test_real_image_script_on_a_data_tree(pea, tmp_path):
    """make_dense_pointset.make_dense_keypoints over real_data/<dataset>/<class>/{images,sparse_keypoints}."""
    from oracle import prune2d
    gv = pea.global_variables
    saved = (gv.processed_image_loc, gv.data_dir)
    try:
        gv.data_dir = str(tmp_path)
        gv.processed_image_loc = str(tmp_path / "real_data")
        rs = np.random.RandomState(8)
        for dataset, stem, cls in (("keypoint-5", "sofa", "sofa"), ("pascal3D", "chair_pascal", "chair")):
            root = tmp_path / "real_data" / dataset / stem
            (root / "images").mkdir(parents=True)
            (root / "sparse_keypoints").mkdir()
            k = 14 if cls == "sofa" else 10
            sparse = {}
            for i in range(7):
                name = "%08d_img" % i
                (root / "images" / (name + ".jpg")).write_bytes(b"")
                kp = np.concatenate([rs.uniform(-10, 240, (k, 2)), (rs.uniform(size=(k, 1)) < 0.8) * 1.0], 1)
                if dataset == "pascal3D" and i == 2:
                    kp[:, 2] = 0
                np.save(root / "sparse_keypoints" / (name + ".npy"), kp)
                sparse[name] = kp
            n = pea.make_dense_pointset.make_dense_keypoints(dataset, cls, "pascal")
            listed = (tmp_path / "image_lists" / "dense_keypoints" / f"{dataset}_{cls}_pascal.txt").read_text().split()
            want = {name: prune2d.make_dense_keypoints_image(kp, cls, dataset) for name, kp in sparse.items()}
            assert sorted(listed) == sorted(name for name, w in want.items() if w is not None) and n == len(listed)
            for name, w in want.items():
                path = root / "dense_keypoints" / (name + ".npy")
                if w is None:
                    assert not path.exists()
                else:
                    got = np.load(path)
                    assert got.dtype == np.float64 and np.array_equal(got, w)
    finally:
        gv.processed_image_loc, gv.data_dir = saved


def test_make_disparity_unfused_with_one_shared_depth_map(pea, golden_disparity):
    """fused=False with a single (H,W) depth map for several renders (the fused path always took it)."""
    z = golden_disparity
    pts, depth, crop = z["cloud_pts"], z["cloud_depth"], z["cloud_crop"]
    params = np.stack([z["cloud_params"], z["cloud_params"] + [3.0, 1.0, 0.0, 0.0]])
    a = pea.disparity.make_disparity_maps(pts, params, depth, [crop, crop], fused=True)
    b = pea.disparity.make_disparity_maps(pts, params, depth, [crop, crop], fused=False)
    assert a.shape == (2, 224, 224, 2) and np.array_equal(a, b) and np.array_equal(a[0], z["cloud_flow"])


def test_flow_splat_slice_semantics(pea):
    uv = np.array([[0, 1, 2, 223, 230, -5, 100, 50], [50, 50, 50, 223, 10, 60, -3, 50], [1.0] * 8])
    uv2 = uv + np.array([[1.5], [-2.0], [0.0]]) * np.arange(1, 9)
    truth = np.array([1, 1, 1, 1, 1, 1, 1, 0], dtype=bool)
    got = pea.disparity.flow_splat(uv, uv2, truth)
    keep = truth
    want = odisp.splat_flow_literal(uv[:, keep], (uv2 - uv)[:, keep])
    assert np.array_equal(got, want)
    assert not got[48:52, 48:52].any()            # the masked point wrote nothing
    empty = pea.disparity.flow_splat(np.zeros((3, 0)), np.zeros((3, 0)), np.zeros(0, bool))
    assert empty.shape == (224, 224, 2) and not empty.any()


@pytest.mark.parametrize("spread,side", [(1, 224), (2, 224), (3, 224), (4, 224), (5, 224), (2, 64), (4, 9), (2, 3),
                                         (2, 1000), (6, 300), (4, 3000)])
def test_flow_splat_wild_coordinates(pea, spread, side):
    """Coordinates far outside the map, on every border and negative (Python slices wrap): the
    anchor-image path (spread <= 4) and the per-pixel atomic path must both match the literal loop."""
    rs = np.random.RandomState(100 * spread + side)
    n = 4000
    uv = np.stack([rs.randint(-2 * side - 9, 2 * side + 9, n), rs.randint(-2 * side - 9, 2 * side + 9, n),
                   np.ones(n)]).astype(np.float64)
    uv[:2, :600] = rs.randint(-2 * spread - 1, 2 * spread + 2, (2, 600))              # around the origin
    uv[:2, 600:1200] = side + rs.randint(-2 * spread - 1, 2 * spread + 2, (2, 600))   # around the far corner
    uv[:2, 1200:1800] = -side + rs.randint(-2 * spread - 1, 2 * spread + 2, (2, 600)) # wraps onto the origin
    uv[:2, 1800:2600] = rs.randint(0, side, (2, 800))                                  # inside
    uv[:2] += rs.uniform(0, 0.99, (2, n)) * np.sign(uv[:2])                            # astype(int) truncates
    uv2 = uv + rs.uniform(-9, 9, (3, n))
    truth = rs.rand(n) > 0.2
    got = pea.disparity.flow_splat(uv, uv2, truth, side=side, spread=spread)
    want = odisp.splat_flow_literal(uv[:, truth], (uv2 - uv)[:, truth], side=side, k=spread)
    assert np.array_equal(got, want)
    assert np.count_nonzero(want) > 0


@pytest.mark.parametrize("spread,h,w", [(2, 540, 960), (2, 960, 540), (1, 7, 300), (3, 448, 447), (5, 33, 1200), (2, 1, 5)])
def test_flow_splat_rectangular_maps(pea, spread, h, w):
    """(H, W) maps as the flow demo writes them (get_of.py:54-60): rows wrap and clip against H, columns against W."""
    rs = np.random.RandomState(7 * h + w + spread)
    n = 5000
    uv = np.stack([rs.randint(-2 * w - 9, 2 * w + 9, n), rs.randint(-2 * h - 9, 2 * h + 9, n),
                   np.ones(n)]).astype(np.float64)
    uv[:2, :600] = rs.randint(-2 * spread - 1, 2 * spread + 2, (2, 600))
    uv[0, 600:1200] = w + rs.randint(-2 * spread - 1, 2 * spread + 2, 600)
    uv[1, 600:1200] = h + rs.randint(-2 * spread - 1, 2 * spread + 2, 600)
    uv[0, 1200:1800] = -w + rs.randint(-2 * spread - 1, 2 * spread + 2, 600)
    uv[1, 1200:1800] = -h + rs.randint(-2 * spread - 1, 2 * spread + 2, 600)
    uv[0, 1800:3200] = rs.randint(0, w, 1400)
    uv[1, 1800:3200] = rs.randint(0, h, 1400)
    uv[:2] += rs.uniform(0, 0.99, (2, n)) * np.sign(uv[:2])
    uv2 = uv + rs.uniform(-9, 9, (3, n))
    truth = rs.rand(n) > 0.2
    got = pea.disparity.flow_splat(np.stack([uv, uv]), np.stack([uv2, uv2]), np.stack([truth, ~truth]), side=(h, w),
                                   spread=spread)
    assert got.shape == (2, h, w, 2)
    for m, keep in enumerate((truth, ~truth)):
        want = odisp.splat_flow_literal(uv[:, keep], (uv2 - uv)[:, keep], side=(h, w), k=spread)
        assert np.array_equal(got[m], want)
    assert np.count_nonzero(got) > 0


@pytest.mark.parametrize("tag", ["cloud", "mesh", "border", "shipped"])
def test_demo_flow_golden(pea, golden_demo_flow, tag):
    """The flow demo (render_wt_pt_proj/get_of.py:38-62) against maps made by the imported reference primitives."""
    from conftest import dense_flow
    z = golden_demo_flow
    want = dense_flow(z, tag)
    got = pea.disparity.demo_flow(z[f"{tag}_pts"], z[f"{tag}_params"], z[f"{tag}_depth"])
    assert got.shape == (540, 960, 2) and got.dtype == np.float64
    assert np.array_equal(got, want)


def test_demo_flow_batch_100k(pea):
    rs = np.random.RandomState(3)
    pts = rs.uniform(-0.4, 0.4, (100000, 3))
    params = np.array([[30.0, 10.0, 0.0, 3.0], [250.0, 25.0, 5.0, 2.0]])
    depths = []
    for p in params:
        d = ozb.zbuffer(og.get_3d_to_2d(pts, p), 540, 960, radius=2)[2]
        d[~np.isfinite(d)] = 20.0
        depths.append(d.astype(np.float32))
    got = pea.disparity.demo_flow(pts, params, np.stack(depths))
    for m in range(2):
        want = odisp.demo_flow_core(pts, params[m].tolist(), depths[m])
        assert np.array_equal(got[m], want) and np.count_nonzero(want) > 20000
    shared = pea.disparity.demo_flow(pts, params, depths[0])
    assert np.array_equal(shared[0], got[0])
    assert np.array_equal(shared[1], odisp.demo_flow_core(pts, params[1].tolist(), depths[0]))
    with pytest.raises(ValueError):
        pea.disparity.demo_flow(pts, params, np.stack(depths + depths[:1]))


def test_prune_pts_reads_the_exr_depth_render_itself(pea):
    """run_proj.prune_pts(pt_array, "<render>.png1.exr") as the reference calls it (run_proj.py:33-46,49-57):
    no .npy cache, no OpenEXR module; the depth image is the crop of the reference's own Blender render."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blender_depth_crop.exr")
    depth = pea.run_proj.get_depthmap_from_exr(path)
    assert depth.shape == (48, 960) and depth.dtype == np.float32
    ys, xs = np.nonzero(depth < 10.0)
    y, x = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])                  # a pixel on the object
    d = float(depth[y, x])
    pr = np.array([[x, x, x, 5.0], [y, y, y, 5.0], [d, d + 0.1, d + 0.2, 10.0]], dtype=np.float64)
    truth, same = pea.run_proj.prune_pts(pr, path)
    assert np.array_equal(truth, og.depth_visibility_array_a(pr, depth)[0].astype(np.float64))
    assert truth.tolist() == [1.0, 1.0, 0.0, 1.0]                           # |z - depth| <= 0.15; background is 10


def test_point_depth_splat(pea, golden_disparity):
    z = golden_disparity
    assert np.array_equal(pea.io_utils.get_pt_depth_map(z["ptdepth_render"]), z["ptdepth_map"])


def test_zbuffer(pea, golden_geometry):
    r = golden_geometry["g3_near_render"]
    for radius in (0, 2):
        key, win, depth = ozb.zbuffer(r, 540, 960, radius=radius)
        got = pea.zbuffer.zbuffer(r, (540, 960), radius=radius)
        assert np.array_equal(got["winner"], win)
        assert np.array_equal(got["depth"], depth)
        assert np.array_equal(got["packed"].view(np.uint64), key)
        vis = pea.zbuffer.visibility(r, got["depth"], relax=0.01)
        assert np.array_equal(vis, ozb.zbuffer_visibility(r, depth, relax=0.01))
        # straight from the packed keys, without the decoded depth image
        assert np.array_equal(pea.zbuffer.visibility(r, got["packed"], relax=0.01), vis)
    # many poses: more than 2^20 blocks in the resolve launch
    import torch
    big = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(r, (600,) + r.shape))).cuda()
    got = pea.zbuffer.zbuffer(big, (540, 960), want=("depth",))
    assert torch.equal(got["depth"][599], got["depth"][0])
    assert np.array_equal(got["depth"][599].cpu().numpy(), ozb.zbuffer(r, 540, 960)[2])
    del big, got
    # batched over poses
    rr = np.stack([golden_geometry["g3_near_render"], golden_geometry["g3_std_render"]])
    got = pea.zbuffer.zbuffer(rr, (540, 960))
    assert np.array_equal(got["winner"][1], ozb.zbuffer(rr[1], 540, 960)[1])


def _zb_oracle(pts, cam, h, w, radius, relax):
    render = cnative.project_fma(pts, cam)
    with np.errstate(invalid="ignore", over="ignore"):
        key, win, depth = ozb.zbuffer(render, h, w, radius=radius)
        vis = ozb.zbuffer_visibility(render, depth, relax=relax)
    return render, key, win, depth, vis


def test_project_zbuffer_tiled_against_oracle(pea):
    """Screen-tiled fused pass (projection + z-buffer + visibility): keys, winners, depths and flags
    bit-exact against the oracle for points in random order (every tile visits every chunk) and in Morton
    order (chunks culled per tile), several image sizes (odd width: scalar stores), splat radii, and
    points that cannot take part (NaN, inf, behind the camera, off screen)."""
    rs = np.random.RandomState(61)
    pts = rs.uniform(-0.45, 0.45, (7000, 3))
    pts[::97] *= 3.0                                   # some far off screen
    pts[5] = [np.nan, 0.1, 0.1]
    pts[6] = [np.inf, 0.1, 0.1]
    pts[7] = [0.0, 1e300, 0.0]
    poses = np.array([[30.0, 10.0, 0.0, 3.0], [200.0, -5.0, 12.0, 1.1], [95.0, 35.0, -8.0, 0.3]])   # last: camera inside the cloud
    cams = pea.pt2render.camera_matrices(poses)
    order = pea.run_proj.spatial_order(np.nan_to_num(pts, nan=0.0, posinf=1.0, neginf=-1.0))
    for (h, w), radius in (((540, 960), 0), ((540, 960), 2), ((131, 77), 1), ((64, 128), 0)):
        cam_set = cams.copy()
        if (h, w) != (540, 960):                       # move the principal point into the small image
            cam_set[:, 0] += cam_set[:, 2] * (w / 2 - 480)
            cam_set[:, 1] += cam_set[:, 2] * (h / 2 - 270)
        for arr in (pts, pts[order]):
            got = pea.zbuffer.project_zbuffer(arr, cam_set, (h, w), radius=radius, relax=0.01,
                                              want=("packed", "winner", "depth", "truth"))
            for p in range(len(poses)):
                render, key, win, depth, vis = _zb_oracle(arr, cam_set[p], h, w, radius, 0.01)
                assert np.array_equal(got["packed"][p].view(np.uint64), key), (h, w, radius, p)
                assert np.array_equal(got["winner"][p], win) and np.array_equal(got["depth"][p], depth)
                assert np.array_equal(got["truth"][p], vis)
            assert got["truth"].sum() > 100
    # the same results, in the order of the given points, when the pass runs on a processing order of its own
    po = pea.disparity.PointOrder(pts)
    plain = pea.zbuffer.project_zbuffer(pts, cams, radius=2, relax=0.01, want=("winner", "depth", "truth"))
    fast = pea.zbuffer.project_zbuffer(pts, cams, radius=2, relax=0.01, want=("winner", "depth", "truth"), point_order=po)
    for key in ("winner", "depth", "truth"):
        assert np.array_equal(plain[key], fast[key]), key
    with pytest.raises(ValueError):
        pea.zbuffer.project_zbuffer(pts[:-1], cams, point_order=po)
    # flags only, one camera, no images written
    only = pea.zbuffer.project_zbuffer(pts, cams[0], want=("truth",))
    assert set(only) == {"truth"} and np.array_equal(only["truth"], _zb_oracle(pts, cams[0], 540, 960, 0, 0.0)[4])
    # the same pass over projected coordinates
    render = pea.pt2render.get_render_location(pts, cams)
    got = pea.zbuffer.project_zbuffer(None, None, radius=2, relax=0.01, want=("packed", "truth"), pt_render=render)
    for p in range(len(poses)):
        _, key, _, _, vis = _zb_oracle(pts, cams[p], 540, 960, 2, 0.01)
        assert np.array_equal(got["packed"][p].view(np.uint64), key) and np.array_equal(got["truth"][p], vis)
    # empty point set: every pixel empty
    got = pea.zbuffer.project_zbuffer(np.zeros((0, 3)), cams[:2], (33, 40), want=("winner", "depth", "truth"))
    assert (got["winner"] == -1).all() and np.isinf(got["depth"]).all() and got["truth"].shape == (2, 0)


def test_zbuffer_config4_pose_sweep(pea):
    """100k points x 36 distinct poses: ispa_project -> ispa_zbuffer -> visibility (global atomics) and the
    fused tiled pass must agree bit for bit on every pose; 3 poses against the oracle; a re-run of both gives
    the same bits (atomics commute: the result does not depend on scheduling)."""
    import torch
    rs = np.random.RandomState(44)
    pts = rs.uniform(-0.4, 0.4, (100000, 3))
    pts = pts[pea.run_proj.spatial_order(pts)]
    poses = np.array([[az, el, 0.0, 3.0] for az in range(0, 360, 30) for el in (-10.0, 10.0, 35.0)])
    cams = pea.pt2render.camera_matrices(poses)
    pts_d = torch.from_numpy(pts).cuda()
    cams_d = torch.from_numpy(cams).cuda()

    def unfused():
        render = pea.pt2render.get_render_location(pts_d, cams_d)
        zb = pea.zbuffer.zbuffer(render, (540, 960), want=())
        return zb["packed"], pea.zbuffer.visibility(render, zb["packed"], relax=0.01)

    def fused():
        out = pea.zbuffer.project_zbuffer(pts_d, cams_d, (540, 960), relax=0.01)
        return out["packed"], out["truth"]
    pa, ta = unfused()
    pb, tb = fused()
    assert torch.equal(pa, pb) and torch.equal(ta, tb)
    pa2, ta2 = unfused()
    pb2, tb2 = fused()
    assert torch.equal(pa, pa2) and torch.equal(ta, ta2) and torch.equal(pb, pb2) and torch.equal(tb, tb2)
    for p in (0, 17, 35):
        _, key, _, _, vis = _zb_oracle(pts, cams[p], 540, 960, 0, 0.01)
        assert np.array_equal(pb[p].cpu().numpy().view(np.uint64), key)
        assert np.array_equal(tb[p].cpu().numpy(), vis)
        assert 20000 < int(vis.sum()) < 100000


def test_project_zbuffer_against_the_blender_render(pea):
    """The fused z-buffer against the one physical depth measurement the reference ships: Blender's render of its
    bed mesh (see test_zbuffer_oracle_against_the_blender_render for what is compared and how approximate it is),
    and bit for bit against the oracle on the same 800k samples."""
    from conftest import physical_zbuffer_agreement, physical_zbuffer_case
    pts, pose, shipped, stats = physical_zbuffer_case()
    pts = pts[pea.run_proj.spatial_order(pts)]
    cam = pea.pt2render.camera_matrices(np.array([pose]))
    got = pea.zbuffer.project_zbuffer(pts, cam, (540, 960), radius=0, relax=0.01, want=("packed", "depth", "truth"))
    iou, med, near = physical_zbuffer_agreement(got["depth"][0], shipped)
    assert iou > 0.97 and med < 0.003 and near > 0.88, (iou, med, near)
    render, key, win, depth, vis = _zb_oracle(pts, cam[0], 540, 960, 0, 0.01)
    assert np.array_equal(got["packed"][0].view(np.uint64), key) and np.array_equal(got["truth"][0], vis)


@pytest.mark.parametrize("n,radius", [(100000, 0), (100000, 2), (150000, 0), (150000, 1)])
def test_project_zbuffer_unsorted_large_models(pea, n, radius):
    """Points in the reference's own (random) sampling order: every chunk of 64 consecutive points spans the whole
    object, so an object tile is hit by all 1563 (100k points) or 2344 (150k) chunks.  That is more than the
    1024 entries of the tile's chunk list: the list is built in several segments, each after the first attempt to
    take four rounds of boxes overflowed and was retried with fewer.  Keys, winners, depths and flags bit-exact
    against the oracle; two runs give the same bits."""
    rs = np.random.RandomState(97 + n // 1000 + radius)
    pts = rs.uniform(-0.4, 0.4, (n, 3))
    pts[::2] *= np.array([1.0, 0.25, 1.0])             # a denser slab, so depths collide
    poses = np.array([[30.0, 10.0, 0.0, 3.0], [250.0, 35.0, 5.0, 2.0]])
    cams = pea.pt2render.camera_matrices(poses)
    want = ("packed", "winner", "depth", "truth")
    got = pea.zbuffer.project_zbuffer(pts, cams, (540, 960), radius=radius, relax=0.01, want=want)
    again = pea.zbuffer.project_zbuffer(pts, cams, (540, 960), radius=radius, relax=0.01, want=want)
    for key in want:
        assert np.array_equal(got[key], again[key]), key
    for p in range(len(poses)):
        render, key, win, depth, vis = _zb_oracle(pts, cams[p], 540, 960, radius, 0.01)
        assert np.array_equal(got["packed"][p].view(np.uint64), key), (n, radius, p)
        assert np.array_equal(got["winner"][p], win) and np.array_equal(got["depth"][p], depth)
        assert np.array_equal(got["truth"][p], vis)
        assert int(vis.sum()) > 5000
    # the same through projected coordinates (chunk boxes from the pixels themselves)
    render = pea.pt2render.get_render_location(pts, cams)
    got2 = pea.zbuffer.project_zbuffer(None, None, radius=radius, relax=0.01, want=("packed", "truth"), pt_render=render)
    assert np.array_equal(got2["packed"], got["packed"]) and np.array_equal(got2["truth"], got["truth"])


# ---------------------------------------------------------------- dense nearest neighbour
def _nn_check(pea, q, r, k, bf16=False, index_base=1, split=False, pad=True):
    import torch
    b = q.shape[0]
    off_grid = q.shape[1] % 32 or int(np.prod(q.shape[2:])) % 8 or int(np.prod(r.shape[2:])) % 8
    if pad and off_grid:
        # shapes off the grid of the fast kernels: also as they are, through the library's guarded loaders
        _nn_check(pea, q, r, k, bf16=bf16, index_base=index_base, split=split, pad=False)
    if split:
        dist, ind, st = pea.correspondence.dense_nn(q, r, k, index_base, return_stats=True, filter="split-always",
                                                    pad=pad)
    elif bf16:
        qd = torch.from_numpy(q).cuda().to(torch.bfloat16)
        rd = torch.from_numpy(r).cuda().to(torch.bfloat16)
        dist, ind, st = pea.correspondence.dense_nn(qd, rd, k, index_base, return_stats=True, pad=pad)
        dist, ind = dist.cpu().numpy(), ind.cpu().numpy()
        q, r = cnative.round_bf16(q), cnative.round_bf16(r)
    else:
        dist, ind, st = pea.correspondence.dense_nn(q, r, k, index_base, return_stats=True, pad=pad)
    assert ind.dtype == np.int32 and dist.dtype == np.float32
    for i in range(b):
        want_i, want_d = cnative.nn_exact(q[i].reshape(q.shape[1], -1), r[i].reshape(r.shape[1], -1), k, index_base)
        assert np.array_equal(ind[i], want_i), f"pair {i}: {np.sum(ind[i] != want_i)} index mismatches"
        finite = np.isfinite(want_d)
        # squared distances: 1e-4 RELATIVE, no absolute floor (an exact match must come back as 0)
        assert np.all(np.abs(dist[i][finite] - want_d[finite]) <= 1e-4 * np.abs(want_d[finite]))
        assert np.all(np.isinf(dist[i][~finite]))
    return st


def _unit(rs, shape):
    x = rs.randn(*shape).astype(np.float32)
    return x / np.sqrt((x ** 2).sum(1, keepdims=True))


def test_nn_fused_l2norm_prologue(pea):
    """dense_nn(normalize=True) = L2NormLayer (l2norm_layer.py:18-29) + KNN in one call: bit-identical to
    l2norm() followed by dense_nn(), and equal to the oracle chain (oracle.correlation.l2norm, then
    cnative.nn_exact on ITS normalised features) wherever the two float32 normalisations agree."""
    rs = np.random.RandomState(71)
    for (b, c, nq, nr, k, flt) in ((2, 128, 1024, 2048, 1, "native"), (1, 33, 130, 517, 5, "native"),
                                   (2, 64, 264, 520, 1, "split")):
        q = (rs.randn(b, c, nq) * rs.uniform(0.2, 5.0, (b, 1, nq))).astype(np.float32)
        r = (rs.randn(b, c, nr) * rs.uniform(0.2, 5.0, (b, 1, nr))).astype(np.float32)
        d0, i0 = pea.correspondence.dense_nn(pea.correspondence.l2norm(q), pea.correspondence.l2norm(r), k, filter=flt)
        d1, i1 = pea.correspondence.dense_nn(q, r, k, filter=flt, normalize=True)
        assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.int32), d1.view(np.int32))
        qn, rn = pea.correspondence.l2norm(q), pea.correspondence.l2norm(r)
        assert np.allclose(qn, ocorr.l2norm(q), rtol=1e-6, atol=1e-7)
        for p in range(b):
            want_i, want_d = cnative.nn_exact(qn[p], rn[p], k, 1)       # exact for the normalised features
            assert np.array_equal(i1[p], want_i)
            assert np.all(np.abs(d1[p] - want_d) <= 1e-4 * np.abs(want_d))
    with pytest.raises(ValueError):
        import torch
        pea.correspondence.dense_nn(torch.zeros(1, 32, 8, device="cuda", dtype=torch.bfloat16),
                                    torch.zeros(1, 32, 8, device="cuda", dtype=torch.bfloat16), normalize=True)


@pytest.mark.parametrize("tag", ["b1c128", "b2c64"])
@pytest.mark.parametrize("k", [1, 5])
def test_nn_against_the_reference_contraction(pea, golden_nn_crosspin, tag, k):
    """dense_nn against numbers the reference produced: the ranking of feature1's cells for every template cell
    by the reference's own all-pairs contraction (eccv_net.py:193-210; tests/golden/nn_crosspin.npz).  float32
    (both filters): the reference's best cell / five best cells in order wherever its float32 values decide them.
    bfloat16: the same wherever the gaps also exceed the bfloat16 rounding of the inputs (2^-7 of the best value)."""
    import torch
    from conftest import crosspin_cases
    z = golden_nn_crosspin
    for q, r, want, decided in crosspin_cases(z, tag, k):
        for flt in ("native", "split-always"):
            dist, ind = pea.correspondence.dense_nn(q[None], r[None], k, 1, filter=flt)
            assert np.array_equal(ind[0][:, decided], want[:, decided]), flt
            # d^2 = 2 - 2 q.r on unit vectors: ascending distances = the reference's descending values
            assert np.all(np.diff(dist[0], axis=0) >= 0)
        # through the C ABI as it is (no padding of the 784-cell maps by the Python layer)
        dist, ind = pea.correspondence.dense_nn(q[None], r[None], k, 1, pad=False)
        assert np.array_equal(ind[0][:, decided], want[:, decided])
    top, val = z[f"{tag}_top6"].astype(np.int64), z[f"{tag}_top6_values"].astype(np.float64)
    f1, f2 = z[f"{tag}_f1"], z[f"{tag}_f2"]
    b, c = f1.shape[:2]
    qd = torch.from_numpy(f2.reshape(b, c, 784)).cuda().to(torch.bfloat16)
    rd = torch.from_numpy(f1).cuda().to(torch.bfloat16)
    dist, ind = pea.correspondence.dense_nn(qd, rd, k, 1)
    ind = ind.cpu().numpy()
    wide = (val[:, :, 0] > 0) & ((val[:, :, :k] - val[:, :, 1:k + 1]) > 2.0 ** -7 * val[:, :, :1]).all(axis=2)
    assert wide.sum() > (100 if k == 1 else 3)
    for i in range(b):
        assert np.array_equal(ind[i][:, wide[i]], (top[i, :, :k].T + 1)[:, wide[i]])


@pytest.mark.parametrize("k", [1, 5, 8])
def test_nn_small_ragged(pea, k):
    rs = np.random.RandomState(k)
    for (b, c, nq, nr) in ((1, 128, 300, 333), (3, 24, 130, 517), (2, 7, 64, 9), (1, 33, 1, 1025)):
        _nn_check(pea, _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr)), k)


def test_nn_empty_and_single(pea):
    rs = np.random.RandomState(2)
    d, i = pea.correspondence.dense_nn(np.zeros((2, 16, 0), np.float32), _unit(rs, (2, 16, 40)), 1)
    assert d.shape == (2, 1, 0) and i.shape == (2, 1, 0)
    _nn_check(pea, _unit(rs, (1, 16, 1)), _unit(rs, (1, 16, 1)), 1)
    _nn_check(pea, _unit(rs, (1, 1, 5)), _unit(rs, (1, 1, 7)), 1)          # one channel: every distance ties or not
    with pytest.raises(ValueError):
        pea.correspondence.dense_nn(_unit(rs, (1, 16, 4)), _unit(rs, (1, 8, 4)), 1)
    with pytest.raises(ValueError):
        pea.correspondence.dense_nn(_unit(rs, (1, 16, 4)), _unit(rs, (1, 16, 4)), 9)


def test_nn_fewer_refs_than_k(pea):
    rs = np.random.RandomState(3)
    _nn_check(pea, _unit(rs, (1, 16, 50)), _unit(rs, (1, 16, 3)), 5)


def test_nn_ties_and_duplicates(pea):
    rs = np.random.RandomState(9)
    q = _unit(rs, (2, 128, 256))
    r = _unit(rs, (2, 128, 512))
    r[:, :, 300] = r[:, :, 17]            # exact duplicate references: lowest index must win
    r[:, :, 450] = r[:, :, 17]
    q[:, :, 5] = r[:, :, 17]
    q[:, :, 6] = r[:, :, 17]
    r[0, :, 100:164] = r[0, :, 0:1]       # a block of 65 identical references
    st = _nn_check(pea, q, r, 1)
    assert st["exact_path_queries"] >= 2   # the duplicates must have gone through the float64 pass
    _nn_check(pea, q, r, 5)
    # degenerate input: every reference identical -> every query ambiguous
    r2 = np.repeat(r[:, :, :1], 200, axis=2)
    st = _nn_check(pea, q, r2, 1)
    assert st["exact_path_queries"] == 2 * 256
    st = _nn_check(pea, q, r2, 1, bf16=True)
    assert st["exact_path_queries"] == 2 * 256
    _nn_check(pea, q, r, 1, bf16=True)
    _nn_check(pea, q, r, 5, bf16=True)


def test_nn_near_ties_below_rounding(pea):
    # references that differ from each other by ~1e-7: the filter cannot order them, the exact pass must
    rs = np.random.RandomState(21)
    q = _unit(rs, (1, 128, 128))
    r = _unit(rs, (1, 128, 256))
    base = r[0, :, 40].copy()
    for t in range(41, 60):
        r[0, :, t] = base
        r[0, t % 128, t] = np.nextafter(base[t % 128], np.float32(2.0))
    q[0, :, 0] = base
    _nn_check(pea, q, r, 1)
    _nn_check(pea, q, r, 5)
    # after bfloat16 rounding the near-duplicates collapse into exact ties: lowest index wins
    _nn_check(pea, q, r, 1, bf16=True)
    _nn_check(pea, q, r, 5, bf16=True)
    # references one bfloat16 ulp apart in one channel
    rb = cnative.round_bf16(r)
    for t in range(41, 60):
        rb[0, :, t] = rb[0, :, 40]
        bits = rb[0, t % 128, t:t + 1].view(np.uint32)
        bits += np.uint32(0x10000)                       # next bfloat16 away from zero
    qb = cnative.round_bf16(q)
    qb[0, :, 0] = rb[0, :, 40]
    _nn_check(pea, qb, rb, 1, bf16=True)


def test_nn_unnormalised_and_index_base(pea):
    rs = np.random.RandomState(4)
    q = (rs.randn(2, 64, 200) * 3.0).astype(np.float32)
    r = (rs.randn(2, 64, 700) * rs.uniform(0.1, 5.0, (1, 1, 700))).astype(np.float32)
    _nn_check(pea, q, r, 1, index_base=0)
    _nn_check(pea, q, r, 5, index_base=0)


def test_nn_nchw_layout_and_one_based(pea):
    rs = np.random.RandomState(6)
    f1 = _unit(rs, (2, 128, 56, 56))
    f2 = _unit(rs, (2, 128, 56, 56))
    dist, ind = pea.correspondence.dense_nn(f1, f2, 1)
    assert ind.shape == (2, 1, 3136) and ind.min() >= 1 and ind.max() <= 3136
    want, _ = cnative.nn_exact(f1[1].reshape(128, -1), f2[1].reshape(128, -1), 1, 1)
    assert np.array_equal(ind[1], want)
    xy = pea.correspondence.ind2coord(ind, 56)
    assert np.array_equal(xy, okp.ind2coord(ind, 56))


def test_nn_single_pair_path_matches_batch_path(pea):
    """A call on up to 16384 queries takes kernels built for latency (more threads per query, fused
    re-scoring and refinement); larger batches take the throughput kernels.  Both must return the same
    bits: one awkward pair (duplicates, near ties, close matches, a self match) alone and as five copies."""
    import torch
    rs = np.random.RandomState(123)
    q = _unit(rs, (1, 128, 4096))
    r = _unit(rs, (1, 128, 4096))
    r[0, :, 3000] = r[0, :, 17]                       # duplicate references: lowest index wins
    r[0, :, 100:140] = r[0, :, 99:100]                # a run of identical references
    q[0, :, 5] = r[0, :, 17]                          # exact self matches (distance 0)
    q[0, :, 6] = r[0, :, 120]
    base = r[0, :, 500].copy()
    for t in range(501, 520):                         # references one float32 ulp apart
        r[0, :, t] = base
        r[0, t % 128, t] = np.nextafter(base[t % 128], np.float32(2.0))
    q[0, :, 7] = base
    close = rs.permutation(4096)[:1500]               # close matches: refined distances
    q[0, :, 1000:2500] = r[0, :, close] .T + 1e-3 * rs.randn(128, 1500).astype(np.float32)
    for bf16 in (False, True):
        st1 = _nn_check(pea, q, r, 1, bf16=bf16)
        assert st1["exact_path_queries"] >= 3
        if bf16:
            qd = torch.from_numpy(q).cuda().to(torch.bfloat16)
            rd = torch.from_numpy(r).cuda().to(torch.bfloat16)
        else:
            qd, rd = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
        d1, i1, s1 = pea.correspondence.dense_nn(qd, rd, 1, return_stats=True)
        d5, i5, s5 = pea.correspondence.dense_nn(qd.expand(5, -1, -1).contiguous(), rd.expand(5, -1, -1).contiguous(), 1,
                                                 return_stats=True)
        for c in range(5):
            assert torch.equal(i5[c], i1[0])
            assert torch.equal(d5[c].view(torch.int32), d1[0].view(torch.int32))
        assert s5["full_rescans"] == 5 * s1["full_rescans"]
        assert s5["candidate_rescores"] == 5 * s1["candidate_rescores"]


def test_nn_full_pair_64x64x128(pea):
    rs = np.random.RandomState(2)
    q = _unit(rs, (1, 128, 64, 64))
    r = _unit(np.random.RandomState(3), (1, 128, 64, 64))
    st = _nn_check(pea, q.reshape(1, 128, -1), r.reshape(1, 128, -1), 1)
    assert st["exact_path_queries"] < 4096 // 8


def test_nn_f32_query_resident_filter_shapes(pea):
    # float32 top-1 with C in {32,64,96,128} and 4-aligned cell counts takes the filter that keeps the
    # query fragments in registers; ragged tiles, a split reference range and a single chunk
    rs = np.random.RandomState(77)
    for (b, c, nq, nr) in ((2, 128, 260, 1028), (1, 96, 132, 516), (3, 64, 4, 2052), (1, 32, 1024, 100), (1, 128, 64, 8192)):
        _nn_check(pea, _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr)), 1)


def test_nn_batch_and_bf16(pea):
    rs = np.random.RandomState(8)
    q = _unit(rs, (6, 128, 1024))
    r = _unit(rs, (6, 128, 1024))
    _nn_check(pea, q, r, 1)
    _nn_check(pea, q, r, 1, bf16=True)
    _nn_check(pea, q[:2], r[:2], 5, bf16=True)


def test_nn_bf16_shapes_and_error_model(pea):
    import torch
    rs = np.random.RandomState(31)
    # native bf16 MFMA path: C in {32,64,96,128}, cell counts multiples of 8 but not of the tile
    for (b, c, nq, nr) in ((2, 64, 264, 520), (1, 32, 8, 136), (3, 96, 512, 384), (1, 128, 2048, 4096)):
        _nn_check(pea, _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr)), 1, bf16=True)
    _nn_check(pea, _unit(rs, (2, 128, 520)), _unit(rs, (2, 128, 1032)), 5, bf16=True)
    # 160 .. 256 channels: the same kernels with one block per CU
    for c in (160, 192, 224, 256):
        _nn_check(pea, _unit(rs, (2, c, 264)), _unit(rs, (2, c, 1032)), 1, bf16=True)
        _nn_check(pea, _unit(rs, (1, c, 520)), _unit(rs, (1, c, 776)), 5, bf16=True)
    _nn_check(pea, _unit(rs, (1, 256, 1024)), _unit(rs, (1, 256, 4096)), 8, bf16=True)
    # shapes the native path does not take (converting loader + float32 MFMA)
    _nn_check(pea, _unit(rs, (2, 48, 100)), _unit(rs, (2, 48, 333)), 1, bf16=True)
    # the margin test relies on a rounding model of the bf16 MFMA: measure the real error of the
    # filter scores (through the returned distances) on benign and on cancellation-heavy data
    for scale_mix in (False, True):
        q = _unit(rs, (2, 128, 1024))
        r = _unit(rs, (2, 128, 2048))
        if scale_mix:       # a few huge channels that cancel: partial sums far above the result
            q[:, :4] *= 30.0
            r[:, :4] *= 30.0
            r[:, 1] = -r[:, 0] * (q[:, 0:1].mean() / max(abs(q[:, 1:2].mean()), 1e-3))
        qd = torch.from_numpy(q).cuda().to(torch.bfloat16)
        rd = torch.from_numpy(r).cuda().to(torch.bfloat16)
        qr, rr = cnative.round_bf16(q), cnative.round_bf16(r)
        # k = 1 runs the 16x16x32 kernel (4 tagged mantissa bits), k = 5 the 32x32x16 kernel (7 tagged bits)
        for k, tag_bits in ((1, 4), (5, 7)):
            dist, ind = pea.correspondence.dense_nn(qd, rd, k)
            for i in range(2):
                _, want_d = cnative.nn_exact(qr[i], rr[i], k, 1)
                qn = np.sqrt((qr[i].astype(np.float64) ** 2).sum(0))
                rmax2 = (rr[i].astype(np.float64) ** 2).sum(0).max()
                scale = qn * np.sqrt(rmax2) + 0.5 * rmax2
                bound = 2.0 * 9 * 2 * 2.0 ** -24 * scale      # the library's per-score MFMA model, no safety
                tag = 2.0 ** (tag_bits - 23) * scale           # the tag replaces the low mantissa bits
                err = np.abs(dist[i].cpu().numpy().astype(np.float64) - want_d) / 2.0
                # the distance also carries the float32 rounding of |q|^2 and of the final subtraction
                slack = 2.0 ** -23 * (qn ** 2 + scale)
                worst = float(((err - tag - slack) / bound).max())
                print(f"bf16 filter error / model bound (k={k}, mix={scale_mix}):", worst)
                assert worst <= 0.5, worst


def test_nn_bf16_257_to_512_channels(pea):
    """bfloat16 maps of more than 256 channels take the kernel with a two-image ring (multiples of 64 channels up
    to 512; other counts are padded inside the call): every k template, ragged tiles, split reference ranges, one
    chunk, the padded counts, and the same counts through the guarded loaders (pad=False)."""
    import torch
    rs = np.random.RandomState(97)
    for c in (320, 384, 448, 512):
        _nn_check(pea, _unit(rs, (2, c, 264)), _unit(rs, (2, c, 1032)), 1, bf16=True)
        _nn_check(pea, _unit(rs, (1, c, 520)), _unit(rs, (1, c, 776)), 5, bf16=True)
    _nn_check(pea, _unit(rs, (1, 512, 1024)), _unit(rs, (1, 512, 4096)), 8, bf16=True)
    _nn_check(pea, _unit(rs, (3, 384, 8)), _unit(rs, (3, 384, 64)), 1, bf16=True)        # one chunk
    _nn_check(pea, _unit(rs, (1, 320, 40)), _unit(rs, (1, 320, 8200)), 5, bf16=True)     # many splits
    for c in (260, 300, 450):                                                            # padded to 320 / 320 / 512
        _nn_check(pea, _unit(rs, (2, c, 100)), _unit(rs, (2, c, 333)), 1, bf16=True)
        _nn_check(pea, _unit(rs, (1, c, 129)), _unit(rs, (1, c, 515)), 5, bf16=True)
    # duplicates and a run of identical references across a chunk boundary: lowest index wins
    q = _unit(rs, (1, 512, 256))
    r = _unit(rs, (1, 512, 1024))
    r[0, :, 60:70] = r[0, :, 59:60]
    r[0, :, 900] = r[0, :, 3]
    q[0, :, 5] = r[0, :, 3]
    q[0, :, 6] = r[0, :, 65]
    _nn_check(pea, q, r, 1, bf16=True)
    _nn_check(pea, q, r, 5, bf16=True)
    # the same maps through the guarded loaders: same indices, same distance bits
    qd = torch.from_numpy(_unit(rs, (2, 300, 264))).cuda().to(torch.bfloat16)
    rd = torch.from_numpy(_unit(rs, (2, 300, 1032))).cuda().to(torch.bfloat16)
    for k in (1, 5):
        d1, i1 = pea.correspondence.dense_nn(qd, rd, k)
        d2, i2 = pea.correspondence.dense_nn(qd, rd, k, pad=False)
        assert torch.equal(i1, i2)
        assert torch.allclose(d1, d2, rtol=1e-4, atol=1e-6)


def test_nn_bf16_against_float32_oracle(pea):
    """bfloat16 features vs the float32 oracle on the unrounded features: wherever the float32 margin
    (d2 - d1) / max(d1, 1e-6) exceeds 2^-7 the nearest neighbour must be the same; distances within
    1e-2 relative (plus 1e-2 absolute for near-zero distances).  The mismatch rate elsewhere is printed."""
    import torch
    rs = np.random.RandomState(53)
    q = _unit(rs, (4, 128, 1024))
    r = _unit(rs, (4, 128, 4096))
    # a slice of easy, well separated matches: noisy copies of references
    q[:, :, :256] = r[:, :, 100:356] + 0.02 * rs.randn(4, 128, 256).astype(np.float32)
    dist, ind = pea.correspondence.dense_nn(torch.from_numpy(q).cuda().to(torch.bfloat16),
                                            torch.from_numpy(r).cuda().to(torch.bfloat16), 1)
    dist, ind = dist.cpu().numpy(), ind.cpu().numpy()
    checked = mismatched = 0
    for i in range(4):
        want_i, want_d = cnative.nn_exact(q[i], r[i], 2, 1)
        margin = (want_d[1] - want_d[0]) / np.maximum(want_d[0], 1e-6)
        sure = margin > 2.0 ** -7
        assert np.array_equal(ind[i, 0][sure], want_i[0][sure])
        checked += int(sure.sum())
        mismatched += int((ind[i, 0][~sure] != want_i[0][~sure]).sum())
        assert np.all(np.abs(dist[i, 0] - want_d[0]) <= 1e-2 * np.abs(want_d[0]) + 1e-2)
    assert checked >= 4 * 256
    print(f"bf16 vs float32 oracle: {checked} of 4096 queries above the 2^-7 margin all agree; "
          f"{mismatched} of the other {4096 - checked} differ")


def test_nn_randomised_shapes_all_paths(pea):
    """Seeded sweep over shapes that select every filter variant (query-resident / LDS-staged float32,
    16x16x32 and 32x32x16 bfloat16 with one or two blocks per CU, split, converting loader), with one
    or several reference splits, ragged tiles and k up to 8, each checked against the oracle."""
    rs = np.random.RandomState(2024)
    channels = [8, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 272]
    cases = 0
    for trial in range(48):
        c = channels[trial % len(channels)]
        b = int(rs.randint(1, 4))
        # multiples of 8 most of the time (fast paths), sometimes of 4 or odd (guarded loaders)
        unit = (8, 8, 8, 4, 1)[int(rs.randint(0, 5))]
        nq = max(unit, int(rs.randint(1, 700)) // unit * unit)
        nr = max(unit, int(rs.randint(1, 3000)) // unit * unit)
        if trial % 6 == 0:
            nq, nr = 8 * int(rs.randint(1, 20)), 8 * int(rs.randint(600, 1200))     # few queries: split references
        k = (1, 1, 5, 8, 2, 1)[trial % 6]
        mode = ("f32", "bf16", "split")[trial % 3]
        q, r = _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr))
        if trial % 5 == 0:                    # a few exact duplicates
            r[:, :, nr // 2] = r[:, :, 0]
            q[:, :, 0] = r[:, :, 0]
        _nn_check(pea, q, r, k, bf16=(mode == "bf16"), split=(mode == "split"), index_base=trial % 2)
        cases += 1
    assert cases == 48


def test_nn_split_filter(pea):
    """float32 inputs through the split-bfloat16 filter: same exact indices as the float32 path."""
    rs = np.random.RandomState(41)
    for (b, c, nq, nr) in ((2, 128, 1024, 2048), (1, 64, 264, 520), (3, 32, 8, 136), (1, 96, 520, 1032)):
        _nn_check(pea, _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr)), 1, split=True)
    # 160 .. 256 channels: one block per CU
    for c in (160, 192, 224, 256):
        _nn_check(pea, _unit(rs, (1, c, 264)), _unit(rs, (1, c, 1032)), 1, split=True)
    # filter="split" on a call too small for the split filter to pay: routed to the float32 filter, same results
    qs, rs_ = _unit(rs, (2, 128, 512)), _unit(rs, (2, 128, 1024))
    d_auto, i_auto = pea.correspondence.dense_nn(qs, rs_, 1, filter="split")
    d_nat, i_nat = pea.correspondence.dense_nn(qs, rs_, 1)
    assert np.array_equal(i_auto, i_nat) and np.array_equal(d_auto, d_nat)
    # requests the split kernel does not take run as plain float32 (k > 1, odd sizes, C = 288)
    _nn_check(pea, _unit(rs, (1, 128, 256)), _unit(rs, (1, 128, 512)), 5, split=True)
    _nn_check(pea, _unit(rs, (2, 48, 100)), _unit(rs, (2, 48, 333)), 1, split=True)
    _nn_check(pea, _unit(rs, (1, 288, 512)), _unit(rs, (1, 288, 1024)), 1, split=True)
    # unnormalised features with a wide dynamic range, duplicates and near-duplicates
    q = (rs.randn(2, 128, 512) * rs.uniform(0.01, 30.0, (1, 128, 1))).astype(np.float32)
    r = (rs.randn(2, 128, 1024) * rs.uniform(0.01, 30.0, (1, 128, 1))).astype(np.float32)
    r[:, :, 700] = r[:, :, 33]
    q[:, :, 9] = r[:, :, 33]
    for t in range(41, 60):
        r[0, :, t] = r[0, :, 40]
        r[0, t % 128, t] = np.nextafter(r[0, t % 128, t], np.float32(100.0))
    q[0, :, 0] = r[0, :, 40]
    st = _nn_check(pea, q, r, 1, split=True)
    assert st["exact_path_queries"] >= 3
    # full pair, self-match property at batch size
    import torch
    g = torch.Generator(device="cuda").manual_seed(9)
    rr = torch.randn(16, 128, 4096, device="cuda", generator=g)
    rr = rr / rr.norm(dim=1, keepdim=True)
    perm = torch.stack([torch.randperm(4096, device="cuda", generator=g) for _ in range(16)])
    qq = torch.gather(rr, 2, perm[:, None, :].expand(-1, 128, -1)).contiguous()
    d, i = pea.correspondence.dense_nn(qq, rr, 1, index_base=0, filter="split-always")
    assert torch.equal(i[:, 0, :].long(), perm) and float(d.abs().max()) == 0.0
    d0, i0 = pea.correspondence.dense_nn(qq, rr.roll(1, 2).contiguous(), 1, index_base=0)
    d1, i1 = pea.correspondence.dense_nn(qq, rr.roll(1, 2).contiguous(), 1, index_base=0, filter="split-always")
    assert torch.equal(i0, i1)


@pytest.mark.parametrize("mode", ["f32", "bf16", "split"])
def test_nn_long_and_uneven_float64_queues(pea, mode):
    """Two pairs whose every query ties exactly among 128 copies of a reference (all 1024 queries of those
    pairs go through the batched float64 pass, more than one round of its workgroups per pair) next to pairs
    with few or no queued queries: lowest index among the ties, other pairs untouched."""
    rs = np.random.RandomState(97)
    b, c, nq, nr = 40, 128, 1024, 1024
    q, r = _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr))
    for heavy in (3, 29):
        base = _unit(rs, (1, c, 8))[0]
        r[heavy] = np.tile(base, (1, nr // 8))
        q[heavy] = base[:, rs.randint(0, 8, nq)]
    st = _nn_check(pea, q, r, 1, bf16=(mode == "bf16"), split=(mode == "split"))
    assert st["full_rescans"] >= 2 * nq


def test_nn_time_grows_with_channels_not_faster(pea):
    """A timing guard with a wide margin: twice the channels must not cost more than five times the time (a
    filter kernel whose register arrays land in scratch memory is a hundred times slower and still exact)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)

    def ms_per_call(c, dtype, flt):
        x = torch.randn(2, 16, c, 4096, device="cuda", generator=g)
        x = (x / x.norm(dim=2, keepdim=True)).to(dtype).contiguous()
        for _ in range(2):
            pea.correspondence.dense_nn(x[0], x[1], 1, filter=flt)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(3):
            pea.correspondence.dense_nn(x[0], x[1], 1, filter=flt)
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / 3

    for dtype, flt in ((torch.bfloat16, "native"), (torch.float32, "split-always"), (torch.float32, "native")):
        base = ms_per_call(128, dtype, flt)
        for c in (192, 224, 256):
            t = ms_per_call(c, dtype, flt)
            assert t < 5.0 * base * c / 128 + 0.2, (str(dtype), flt, c, t, base)


def test_nn_query_counts_off_the_8_cell_grid(pea):
    """Query counts that are not multiples of 8 against an aligned reference map (n sampled correspondences, as
    the KNN layer gets them): the host layer repeats the last query column up to the next multiple of 8 and drops
    the copies; results and shapes are those of the given queries."""
    rs = np.random.RandomState(31)
    for (b, c, nq, nr, k, mode) in ((3, 128, 999, 1024, 1, "f32"), (2, 128, 1001, 1024, 1, "bf16"),
                                    (2, 64, 77, 520, 5, "bf16"), (1, 128, 13, 784, 8, "f32"), (2, 128, 9, 1024, 1, "split")):
        q, r = _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr))
        _nn_check(pea, q, r, k, bf16=(mode == "bf16"), split=(mode == "split"))
    # top-1 against reference maps off the grid (55x55, 63x63 cells): copies of the last cell pad the map and tie
    # with it; queries whose nearest cell IS the last one must still get its index
    for (b, c, nq, nr, mode) in ((2, 128, 512, 3025, "f32"), (2, 128, 999, 3969, "bf16"), (1, 64, 300, 1001, "split")):
        q, r = _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr))
        q[:, :, 5] = r[:, :, nr - 1]                     # exact match with the last cell
        q[:, :, 6] = r[:, :, nr - 1] + 1e-3 * q[:, :, 6]  # and a near match
        _nn_check(pea, q, r, 1, bf16=(mode == "bf16"), split=(mode == "split"))
    # with the layer's query counts on top
    q, r = _unit(rs, (2, 128, 203)), _unit(rs, (2, 128, 1024))
    d, i = pea.correspondence.dense_nn(q, r, 1, num_query=[203, 57])
    d0, i0 = pea.correspondence.dense_nn(q[:, :, :200], r, 1)
    assert d.shape == (2, 1, 203) and np.array_equal(i[:, :, :57], i0[:, :, :57]) and np.all(i[1, :, 57:] == -1)
    assert np.all(np.isinf(d[1, :, 57:])) and np.array_equal(d[0, :, :200], d0[0])


def test_nn_pointers_off_the_16_byte_grid(pea):
    """Feature maps that start one or a few elements into an allocation (views into a larger
    buffer): the 16-byte loaders and LDS-direct paths must stand aside, results unchanged."""
    import torch
    rs = np.random.RandomState(77)
    for (b, c, nq, nr, k) in ((2, 128, 512, 1024, 1), (1, 64, 264, 520, 5), (1, 160, 256, 512, 1)):
        q, r = _unit(rs, (b, c, nq)), _unit(rs, (b, c, nr))
        for mode in ("f32", "bf16", "split"):
            if mode == "split" and k != 1:
                continue
            tdt = torch.bfloat16 if mode == "bf16" else torch.float32
            for off in (1, 2, 3):
                qbuf = torch.empty(q.size + off, dtype=tdt, device="cuda")
                rbuf = torch.empty(r.size + off, dtype=tdt, device="cuda")
                qd, rd = qbuf[off:].view(b, c, nq), rbuf[off:].view(b, c, nr)
                qd.copy_(torch.from_numpy(q))
                rd.copy_(torch.from_numpy(r))
                assert qd.data_ptr() % 16 != 0 and rd.data_ptr() % 16 != 0
                d, i = pea.correspondence.dense_nn(qd, rd, k, 1, filter="split-always" if mode == "split" else "native")
                qo = cnative.round_bf16(q) if mode == "bf16" else q
                ro = cnative.round_bf16(r) if mode == "bf16" else r
                for p in range(b):
                    want_i, _ = cnative.nn_exact(qo[p], ro[p], k, 1)
                    assert np.array_equal(i[p].cpu().numpy(), want_i), (mode, off, c)


def test_nn_topk_contenders_spread_over_many_streams(pea):
    """Few queries against a long reference map (many reference splits), with copies of one reference
    scattered over the map: the exact ties sit in many partial streams and are settled by the
    wave-per-query top-k re-scoring (more candidate slots than lanes for k = 8)."""
    rs = np.random.RandomState(55)
    for c, nq, nr in ((64, 16, 8192), (128, 8, 16384)):
        q, r = _unit(rs, (2, c, nq)), _unit(rs, (2, c, nr))
        spots = rs.choice(nr, 12, replace=False)
        r[:, :, spots] = r[:, :, spots[:1]]
        q[:, :, 0] = r[:, :, spots[0]]
        q[:, :, 1] = r[:, :, spots[0]]
        near = rs.choice(nr, 6, replace=False)             # and a cluster of near (not exact) ties
        for t, j in enumerate(near):
            r[0, :, j] = r[0, :, near[0]]
            r[0, t, j] = np.nextafter(r[0, t, j], np.float32(2.0))
        q[0, :, 2] = r[0, :, near[0]]
        for k in (5, 8, 2):
            for bf16 in (False, True):
                st = _nn_check(pea, q, r, k, bf16=bf16)
                assert st["candidate_rescores"] >= 2, (c, k, bf16, st)


def test_nn_full_rescans_on_long_reference_maps(pea):
    """Long maps run the float64 pass with 1024-thread blocks: a run of identical references longer
    than any kept list forces full rescans (aligned and odd map sizes, k = 1, 5, 8)."""
    rs = np.random.RandomState(56)
    for nr in (8192, 8197):
        q, r = _unit(rs, (1, 64, 24)), _unit(rs, (1, 64, nr))
        r[0, :, 4000:4100] = r[0, :, 4000:4001]
        q[0, :, 0] = r[0, :, 4000]
        q[0, :, 5] = r[0, :, 4000]
        for k in (1, 5, 8):
            for bf16 in (False, True):
                st = _nn_check(pea, q, r, k, bf16=bf16)
                assert st["full_rescans"] >= 2, (nr, k, bf16, st)


def test_nn_top5_c256(pea):
    rs = np.random.RandomState(12)
    _nn_check(pea, _unit(rs, (1, 256, 2048)), _unit(rs, (1, 256, 4096)), 5)


def test_nn_full_size_properties(pea):
    """BASELINE sizes the oracle cannot score in seconds: size-independent properties."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    r = torch.randn(8, 128, 4096, device="cuda", generator=g)
    r = r / r.norm(dim=1, keepdim=True)
    perm = torch.stack([torch.randperm(4096, device="cuda", generator=g) for _ in range(8)])
    q = torch.gather(r, 2, perm[:, None, :].expand(-1, 128, -1)).contiguous()   # queries = permuted refs
    dist, ind = pea.correspondence.dense_nn(q, r, 1, index_base=0)
    assert torch.equal(ind[:, 0, :].long(), perm)            # every query finds itself
    assert float(dist.abs().max()) == 0.0                    # ... at distance exactly 0
    d5, i5 = pea.correspondence.dense_nn(q, r, 5, index_base=0)
    assert torch.equal(i5[:, 0, :].long(), perm) and float(d5[:, 0, :].abs().max()) == 0.0
    assert bool((d5[:, 1:, :] >= d5[:, :-1, :]).all())       # sorted ascending


def test_nn_config3_bf16_one_gpu_share(pea):
    """BASELINE config 3 at one GPU's share: 512 pairs of 64x64x128 bfloat16 features in one call.
    Queries are a permutation of the references (every query must find itself at distance exactly 0),
    except in 4 pairs, which hold independent features and are checked against the oracle on the rounded
    inputs (indices exact, distances 1e-4 relative)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(33)
    b, c, n = 512, 128, 4096
    r = torch.randn(b, c, n, device="cuda", generator=g)
    r = (r / r.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    perm = torch.stack([torch.randperm(n, device="cuda", generator=g) for _ in range(b)])
    q = torch.gather(r, 2, perm[:, None, :].expand(-1, c, -1)).contiguous()
    free = [0, 171, 340, 511]
    fresh = torch.randn(len(free), c, n, device="cuda", generator=g)
    q[free] = (fresh / fresh.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    dist, ind, st = pea.correspondence.dense_nn(q, r, 1, index_base=0, return_stats=True)
    same = torch.ones(b, dtype=torch.bool, device="cuda")
    same[free] = False
    assert torch.equal(ind[same, 0, :].long(), perm[same])
    assert float(dist[same].abs().max()) == 0.0
    for i in free:
        want_i, want_d = cnative.nn_exact(q[i].float().cpu().numpy(), r[i].float().cpu().numpy(), 1, 0)
        assert np.array_equal(ind[i].cpu().numpy(), want_i)
        assert np.all(np.abs(dist[i].cpu().numpy() - want_d) <= 1e-4 * np.abs(want_d))
    # a second run gives the same bits (the partial lists are merged in a fixed order)
    dist2, ind2 = pea.correspondence.dense_nn(q, r, 1, index_base=0)
    assert torch.equal(ind, ind2) and torch.equal(dist.view(torch.int32), dist2.view(torch.int32))


def test_nn_config5_full_size_properties(pea):
    """3 template views x (128x128 grid, 256-d), top-5: the largest BASELINE configuration."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(11)
    r = torch.randn(3, 256, 128 * 128, device="cuda", generator=g)
    r = r / r.norm(dim=1, keepdim=True)
    perm = torch.stack([torch.randperm(128 * 128, device="cuda", generator=g) for _ in range(3)])
    q = torch.gather(r, 2, perm[:, None, :].expand(-1, 256, -1)).contiguous()
    for dt in (torch.float32, torch.bfloat16):
        d5, i5 = pea.correspondence.dense_nn(q.to(dt), r.to(dt), 5, index_base=0)
        assert torch.equal(i5[:, 0, :].long(), perm)               # every query finds itself first
        assert float(d5[:, 0, :].abs().max()) == 0.0
        assert bool((d5[:, 1:, :] >= d5[:, :-1, :]).all())         # sorted ascending
        assert bool((i5 >= 0).all()) and bool((i5 < 128 * 128).all())
        # spot-check three queries of view 1 against the oracle
        qs = q[1].to(dt).float().cpu().numpy()[:, :3]
        want_i, _ = cnative.nn_exact(qs, r[1].to(dt).float().cpu().numpy(), 5, 0)
        assert np.array_equal(i5[1, :, :3].cpu().numpy(), want_i)


def test_projection_config4_pose_sweep(pea):
    """100k points x 360 poses through the fused kernel: spot poses against the bit-exact twin."""
    rs = np.random.RandomState(4)
    pts = rs.uniform(-0.4, 0.4, (100000, 3))
    poses = np.array([[az, 10.0, 0.0, 3.0] for az in range(360)], dtype=np.float64)
    cams = pea.pt2render.camera_matrices(poses)
    depth = np.full((540, 960), 3.0, np.float32)
    truth, out = pea.run_proj.project_prune(pts, cams, depth, [100, 400, 300, 700])
    assert out.shape == (360, 3, 100000)
    for p in (0, 123, 359):
        r = cnative.project_fma(pts, cams[p])
        vis = og.depth_visibility_array_a(r, depth)[0]
        ct, rc = og.crop_visibility_array(r, [100, 400, 300, 700])
        assert np.array_equal(truth[p], vis & ct) and np.array_equal(out[p], rc)
    # odd point count takes the scalar kernels
    t2, o2 = pea.run_proj.project_prune(pts[:99999], cams[:2], depth, [100, 400, 300, 700])
    assert np.array_equal(o2[1], out[1][:, :99999]) and np.array_equal(t2[1], truth[1][:99999])


# ---------------------------------------------------------------- consumers: nearest view, PCK
def test_nearest_view_matches_numpy(pea):
    import torch
    rs = np.random.RandomState(17)
    stored = np.stack([rs.uniform(0, 360, 8000), rs.uniform(-20, 40, 8000), rs.uniform(-30, 30, 8000)], 1)
    stored[4000] = stored[77]                         # duplicates: the first one wins
    stored[7999] = stored[77]
    views = np.stack([rs.uniform(-10, 370, 600), rs.uniform(-30, 50, 600), rs.uniform(-40, 40, 600)], 1)
    views[0] = stored[77]                             # exact hit on the duplicated pose
    views[1] = [359.5, 0, 0]                          # wraps past 360
    views[2] = stored[5] + [180.0, 0, 0]              # difference exactly -180: not wrapped
    views[3] = stored[6] - [180.0, 0, 0]              # difference exactly +180: wrapped
    views[4] = [0, 95, 185]                           # elevation and tilt wrap
    got, dist = pea.correspondence.nearest_view(views, stored, return_distance=True)
    for v in range(len(views)):
        assert got[v] == okp.nearest_view(views[v], stored), v
        d = okp.normalize_pose_delta(stored - views[v])
        d[:, 1:] /= 8.0
        assert dist[v] == np.linalg.norm(d, axis=1).min()        # bit-exact float64
    assert got[0] == 77
    # single pose in, int out; a table already on the device is used in place
    table = torch.from_numpy(stored).cuda()
    one = pea.correspondence.nearest_view([120, 10, 0], table)
    assert isinstance(one, int) and one == okp.nearest_view([120, 10, 0], stored)
    from pose_estimation_amd._abi import IspaError
    with pytest.raises(IspaError):
        pea.correspondence.nearest_view([0, 0, 0], np.zeros((0, 3)))


def test_extract_feature_then_knn(pea):
    from oracle import nn as onn
    rs = np.random.RandomState(29)
    f1 = _unit(rs, (3, 128, 56, 56))
    f2 = _unit(rs, (3, 128, 56, 56))
    corr = np.concatenate([rs.uniform(-6, 230, (3, 300, 4)), np.ones((3, 300, 1))], axis=2).astype(np.float32)
    corr[0, 0, :2] = [0.0, 0.0]
    corr[0, 1, :2] = [223.0, 223.0]
    corr[0, 2, :2] = [223.9, 4.0]                     # x1 clamps at the right edge
    corr[0, 3, :2] = [8.0, 12.0]                      # exactly on a cell
    for mode in ("nearest", "bilinear"):
        got = pea.correspondence.extract_feature(f1, corr, mode=mode)
        assert got.shape == (3, 128, 300) and got.dtype == np.float32
        assert np.array_equal(got, onn.extract_feature(f1, corr, mode=mode)), mode      # bit-exact float32
    # other column layout, as num_coord_elements / coord_start_ind allow
    got = pea.correspondence.extract_feature(f2, corr, coord_start_ind=2, mode="bilinear")
    assert np.array_equal(got, onn.extract_feature(f2, corr, coord_start_ind=2, mode="bilinear"))
    # the layer pair of train.prototxt: extract queries from image 1, match them in image 2
    qf = pea.correspondence.extract_feature(f1, corr)
    _nn_check(pea, qf, f2.reshape(3, 128, -1), 1)
    with pytest.raises(ValueError):
        pea.correspondence.extract_feature(f1, corr[:, :, :1])
    # num_coord bottom of the KNN layer (use_all_query: false): padded query slots are marked
    d_all, i_all = pea.correspondence.dense_nn(qf, f2, 1)
    d_n, i_n = pea.correspondence.dense_nn(qf, f2, 1, num_query=np.array([[300], [17], [0]]))
    assert np.array_equal(i_n[0], i_all[0]) and np.array_equal(i_n[1, :, :17], i_all[1, :, :17])
    assert (i_n[1, :, 17:] == -1).all() and np.isinf(d_n[1, :, 17:]).all() and (i_n[2] == -1).all()


def test_pck_counts_match_numpy(pea):
    import torch
    rs = np.random.RandomState(23)
    b, k, n, width = 3, 2, 700, 56
    ind = rs.randint(1, width * width + 1, (b, k, n)).astype(np.int32)
    corr = np.concatenate([rs.uniform(0, 224, (b, n, 4)), rs.choice([1.0, 0.0, -1.0], (b, n, 1))], axis=2)
    xy = okp.ind2coord(ind[:, 0, :], width)
    corr[:, :40, 2:4] = xy[:, :40] + rs.uniform(-12, 12, (b, 40, 2))     # near-hits around the radii
    corr[:, 40:50, 2:4] = xy[:, 40:50] + [6.0, 8.0]                         # distance exactly 10: not < 10
    corr[:, 50:60, 2:4] = xy[:, 50:60]                                      # distance 0
    corr[:, :60, 4] = 1.0
    radii = [10, 20, 40]
    want = okp.pck_counts(ind, corr, width, radii)
    got = pea.correspondence.pck(ind, corr, width, radii)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    assert want[0].min() >= 10                                              # the planted cases are in play
    # zero-based indices, device tensors in -> device tensor out, float32 correspondences
    got0 = pea.correspondence.pck(torch.from_numpy(ind - 1).cuda(), torch.from_numpy(corr.astype(np.float32)).cuda(),
                                  width, radii, index_base=0)
    assert torch.is_tensor(got0)
    assert np.array_equal(got0.cpu().numpy(), okp.pck_counts(ind, corr.astype(np.float32), width, radii))
    # straight from dense_nn
    q, r = _unit(rs, (2, 32, 56 * 56)), _unit(rs, (2, 32, 56 * 56))
    dist, nn = pea.correspondence.dense_nn(q, r, 1)
    c2 = np.concatenate([rs.uniform(0, 224, (2, 56 * 56, 4)), np.ones((2, 56 * 56, 1))], axis=2)
    assert np.array_equal(pea.correspondence.pck(nn, c2, 56, [30, 60]), okp.pck_counts(nn, c2, 56, [30, 60]))
    with pytest.raises(ValueError):
        pea.correspondence.pck(ind, corr, width, list(range(1, 10)))


def test_hard_negatives_match_numpy(pea):
    import torch
    rs = np.random.RandomState(29)
    b, k, n, width = 3, 2, 900, 56
    ind = rs.randint(1, width * width + 1, (b, k, n)).astype(np.int32)
    corr = np.concatenate([rs.uniform(0, 224, (b, n, 4)), rs.choice([1.0, 0.0, -1.0], (b, n, 1))], axis=2)
    cell = okp.ind2coord(ind[:, 0, :], width, 1)                            # feature coordinates of the 1-NN
    corr[:, :60, 2:4] = (cell[:, :60] + rs.uniform(-6, 6, (b, 60, 2))) * 4  # around the threshold
    corr[:, 60:70, 2:4] = (cell[:, 60:70] + [0.0, 4.0]) * 4                 # distance exactly 4: not > 4
    corr[:, 70:80, 2:4] = cell[:, 70:80] * 4                                # distance 0
    corr[:, :80, 4] = 1.0
    want_mask, want_xy = okp.hard_negatives(ind, corr, width)
    mask, xy = pea.correspondence.hard_negatives(ind, corr, width)
    assert mask.dtype == np.bool_ and np.array_equal(mask, want_mask) and np.array_equal(xy, want_xy)
    assert 10 < want_mask[:, :60].sum() < 170 and not want_mask[:, 60:80].any()
    assert not mask[corr[:, :, 4] != 1].any()
    # zero-based device tensors, another threshold and stride
    m0, xy0 = pea.correspondence.hard_negatives(torch.from_numpy(ind - 1).cuda(), torch.from_numpy(corr).cuda(), width,
                                                thresh=2.5, downsampling_factor=8, index_base=0)
    wm, wxy = okp.hard_negatives(ind, corr, width, thresh=2.5, downsampling=8)
    assert torch.is_tensor(m0) and np.array_equal(m0.cpu().numpy(), wm) and np.array_equal(xy0.cpu().numpy(), wxy)
    # straight from dense_nn; the layer's hard_neg_inds = mask[flag == 1]
    q, r = _unit(rs, (2, 32, 56 * 56)), _unit(rs, (2, 32, 56 * 56))
    _, nn = pea.correspondence.dense_nn(q, r, 1)
    c2 = np.concatenate([rs.uniform(0, 224, (2, 56 * 56, 4)), np.ones((2, 56 * 56, 1))], axis=2)
    got = pea.correspondence.hard_negatives(nn, c2, 56)
    want = okp.hard_negatives(nn, c2, 56)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and want[0].mean() > 0.9
    with pytest.raises(ValueError):
        pea.correspondence.hard_negatives(ind, corr[:, :-1], width)


# ---------------------------------------------------------------- correlation maps
def test_corr_eccv_views_equals_the_per_view_loop(pea):
    rs = np.random.RandomState(61)
    fi = rs.randn(2, 3, 64, 56, 56).astype(np.float32)
    fj = rs.randn(2, 3, 64, 56, 56).astype(np.float32)
    knn = rs.randint(0, 28, (3, 10, 2))
    layer = pea.correspondence.CorrelationMapLayer()
    got = layer.forward_views(fi, fj, knn)
    assert got.shape == (2, 3, 10, 28, 28)
    for v in range(3):                                   # eccv_net.py:282-289: loop, stack, permute
        assert np.array_equal(got[:, v], layer.forward(fi[:, v], fj[:, v], knn[v]))
    with pytest.raises(IndexError):
        layer.forward_views(fi, fj, knn + 20)


@pytest.mark.parametrize("tag", ["b1k1", "b2k10", "b3k14"])
def test_corr_eccv_golden(pea, golden_correlation, tag):
    z = golden_correlation
    f1 = z[f"{tag}_f1"].astype(np.float32)
    f2 = z[f"{tag}_f2"].astype(np.float32)
    layer = pea.correspondence.CorrelationMapLayer()
    got = layer.forward(f1, f2, z[f"{tag}_knn"])
    want = z[f"{tag}_out"]
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.allclose(got, want, rtol=1e-4, atol=1e-7)       # 1e-4 relative (float32 bar)
    with pytest.raises(IndexError):
        layer.forward(f1, f2, np.array([[28, 0]]))


def test_corr_eccv_large_grid(pea):
    """config-5 analogue: K = 10 keypoint maps on a 128 x 128 grid of 256-d features."""
    rs = np.random.RandomState(13)
    f1 = _unit(rs, (1, 256, 128, 128))
    f2 = _unit(rs, (1, 256, 128, 128))
    knn = rs.randint(0, 128, (10, 2))
    got = pea.correspondence.CorrelationMapLayer(grid=128).forward(f1, f2, knn)
    want = ocorr.correlation_map_eccv(f1, f2, knn, grid=128)
    assert got.shape == (1, 10, 128, 128)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-9)


def test_corr_eccv_tiled_and_single_block_paths_agree(pea):
    """Up to 16 keypoints take the tiled kernel (a cell's correlation = four partial chains over channel
    quarters), more take one block per keypoint (one chain): the maps agree to float32 rounding; plus
    resampled inputs, a channel count off the unroll, several views, against the oracle."""
    rs = np.random.RandomState(14)
    f1 = _unit(rs, (2, 37, 96, 96))
    f2 = _unit(rs, (2, 37, 96, 96))
    knn17 = rs.randint(0, 48, (17, 2))
    layer = pea.correspondence.CorrelationMapLayer(grid=48)              # 2304 cells, resampled from 96 x 96
    single = layer.forward(f1, f2, knn17)                                 # 17 keypoints: one block per keypoint
    tiled = layer.forward(f1, f2, knn17[:16])
    assert np.allclose(single[:, :16], tiled, rtol=1e-5, atol=1e-9)
    assert np.allclose(tiled, ocorr.correlation_map_eccv(f1, f2, knn17[:16], grid=48), rtol=1e-4, atol=1e-9)
    assert np.allclose(single, ocorr.correlation_map_eccv(f1, f2, knn17, grid=48), rtol=1e-4, atol=1e-9)
    tiled3 = layer.forward(f1, f2, knn17[:3])
    assert np.array_equal(tiled[:, :3], tiled3)                           # the same kernel, fewer keypoints
    fi, fj = _unit(rs, (2, 3, 64, 64, 64)), _unit(rs, (2, 3, 64, 64, 64))
    knn = rs.randint(0, 64, (3, 10, 2))
    got = pea.correspondence.CorrelationMapLayer(grid=64).forward_views(fi, fj, knn)
    for v in range(3):
        want = ocorr.correlation_map_eccv(fi[:, v], fj[:, v], knn[v], grid=64)
        assert np.allclose(got[:, v], want, rtol=1e-4, atol=1e-9)


def test_resize_and_l2norm(pea, golden_correlation):
    z = golden_correlation
    got = pea.correspondence.resize_bilinear(z["resize_in"], (28, 28))
    assert np.allclose(got, z["resize_out"], rtol=1e-5, atol=1e-6)
    x = np.random.RandomState(1).randn(3, 128, 56, 56).astype(np.float32)
    assert np.allclose(pea.correspondence.l2norm(x), ocorr.l2norm(x), rtol=1e-6, atol=1e-7)


def test_corr_acmm(pea, golden_correlation):
    z = golden_correlation
    rs = np.random.RandomState(2)
    f1 = z["b2k10_f1"].astype(np.float32)
    f2 = z["b2k10_f2"].astype(np.float32)
    mask = (rs.rand(2, 1, 56, 56) > 0.4).astype(np.float32)
    got = pea.correspondence.correlation_map_acmm(f1, f2, mask)
    want = ocorr.correlation_map_acmm(f1, f2, mask)
    assert got.shape == (2, 784, 28, 28)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-6)
    # an all-zero column normalises to NaN in the reference (no epsilon): reproduce it
    f2z = f2.copy()
    f2z[:, :, :4, :4] = 0
    got = pea.correspondence.correlation_map_acmm(f1, f2z, np.ones_like(mask))
    assert np.isnan(got[:, :, 0, 0]).all() and np.isfinite(got[:, :, 20, 20]).all()


@pytest.mark.parametrize("tag", ["b1c128", "b2c32"])
def test_corr_acmm_golden(pea, golden_correlation_acmm, tag):
    """Against net_final_old.py:33-61 itself (run on CPU by tests/golden/make_golden.py), NaN column included."""
    z = golden_correlation_acmm
    f1 = z[f"{tag}_f1"].astype(np.float32)
    f2 = z[f"{tag}_f2"].astype(np.float32)
    got = pea.correspondence.correlation_map_acmm(f1, f2, z[f"{tag}_mask"].astype(np.float32))
    want = z[f"{tag}_out"]
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got[-1, :, 0, 0]).all()
    # 1e-4 relative (float32 bar); entries near the relu threshold are cancellation residues of a 128-term
    # signed sum whose summation-order error is absolute (1e-7 on maps whose entries are <= 1)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-6, equal_nan=True)


def test_tables_on_the_device_side_equal_the_reference(pea, golden_tables):
    t = golden_tables
    dp = pea.dense_pointset
    assert set(dp.skeleton_map) == set(t["skeleton_map"])
    for cls, (bones, tags, seat) in t["skeleton_map"].items():
        o = dp.skeleton_map[cls]
        assert [list(b) for b in o[0]] == bones and list(o[1]) == tags and list(o[2]) == seat
    assert {k: list(v) for k, v in dp.leg_range.items()} == t["leg_range"]
    assert {k: list(v) for k, v in pea.global_variables.syn_template.items()} == t["syn_template"]
    # the class the tables used to lack: 36 anchors, 20 bones -> 200 dense points
    anchors = np.random.RandomState(4).uniform(-0.3, 0.3, (36, 3))
    dense = dp.dense_skeleton(dp.get_additional_coords_keypoint_5(anchors.T, "car"), "car")
    assert dense.shape == (3, 200) and np.array_equal(dense, okp.dense_skeleton_3d(anchors, "car"))


def test_dense_keypoints(pea, golden_geometry, golden_disparity):
    anchors = golden_geometry["g2_anchor_chair"]
    depth = golden_disparity["rb_chair_depth"]
    params = [120.0, 10.0, 0.0, 2.2]
    crop = [60, 480, 250, 720]
    got = pea.dense_pointset.dense_keypoints_for_renders(anchors, "chair", [params], depth, [crop])
    want = okp.make_dense_keypoints_core(anchors, "chair", params, depth, crop)
    assert got.shape == (1, 2, 130) and np.array_equal(got[0], want)


def test_real_image_keypoint_pruning(pea):
    """make_dense_pointset.py:43-99 over batches of synthetic annotations, every class and both
    dataset modes, against the sequential restatement; then the three passes one by one."""
    from oracle import prune2d
    rs = np.random.RandomState(31)
    n_sparse = {"chair": 10, "swivelchair": 13, "bed": 10, "table": 8, "diningtable": 12, "sofa": 14, "car": 36}
    total_pruned = 0
    for cls, k in n_sparse.items():
        for dataset in ("keypoint-5", "pascal3D"):
            if dataset == "pascal3D":
                if cls == "table":
                    continue                     # the pascal extender has no 'table' branch: too few anchors
                k_eff = max(b for bone in pea.dense_pointset.skeleton_map[cls][0] for b in bone) + 1 - \
                    pea.dense_pointset._pascal_extra.get(cls, 0)
            else:
                if cls == "diningtable":
                    continue                     # keypoint-5 has no midpoints for it: the bones index past K
                k_eff = k
            batch = 24
            # objects of a plausible size: a box of 60..200 px somewhere in the image, some annotations
            # outside the image (clipped by the pipeline) and, for pascal, some missing
            centre = rs.uniform(60, 160, (batch, 1, 2))
            sparse = np.concatenate([centre + rs.uniform(-100, 100, (batch, k_eff, 2)),
                                     (rs.uniform(size=(batch, k_eff, 1)) < 0.85).astype(np.float64)], 2)
            if dataset == "keypoint-5":
                sparse[:, :, 2] = 1
            sparse[:3, :, :2] = np.round(sparse[:3, :, :2])           # integer annotations
            sparse[3, :, 0] = sparse[3, 0, 0]                          # all on one vertical line: faces neither way
            if dataset == "pascal3D":
                sparse[4, :, 2] = 0                                    # nothing annotated: not saved
            got, saved = pea.dense_pointset.make_dense_keypoints(sparse, cls, dataset)
            for b in range(batch):
                want = prune2d.make_dense_keypoints_image(sparse[b], cls, dataset)
                assert saved[b] == (want is not None)
                if want is None:
                    assert not got[b].any()
                    continue
                assert np.array_equal(got[b], want), (cls, dataset, b)
                skeleton, info, seat = pea.dense_pointset.skeleton_map[cls]
                total_pruned += int(np.sum((got[b] == 0).all(0)))
    assert total_pruned > 500                    # the passes do fire on these inputs
    # the reference's three entry points, one image at a time, in place
    skeleton, info, seat = pea.dense_pointset.skeleton_map["sofa"]
    for trial in range(6):
        sparse = np.concatenate([rs.uniform(20, 200, (14, 2)), np.ones((14, 1))], 1)
        img = pea.dense_pointset.get_additional_coords_keypoint_5(np.clip(sparse, 0, 223).astype(int).T, "sofa")
        pts = pea.dense_pointset.dense_skeleton(img, "sofa", dim=2)
        a, b = pts.copy(), pts.copy()
        assert pea.dense_pointset.surface_pruned(a, img, seat, pea.dense_pointset.leg_range["sofa"]) is a
        prune2d.surface_pruned(b, img, seat, prune2d.LEG_RANGE["sofa"])
        assert np.array_equal(a, b)
        pea.dense_pointset.dir_pruned(a, img, skeleton, seat, info)
        prune2d.dir_pruned(b, img, skeleton, seat, info)
        assert np.array_equal(a, b)
        pea.dense_pointset.surface_dir_prune(a, img, skeleton, info, seat)
        prune2d.surface_dir_prune(b, img, skeleton, info, seat)
        assert np.array_equal(a, b)
    # points exactly on outline edges and corners, and the band rim, through the C ABI
    import torch
    coords = np.array([[[15.0, 10.0, 20.0, 20.000001, 5.0, 55.0, 55.0, 48.0, 47.0, -1.0]],
                       [[15.0, 12.5, 20.0, 15.0, 10.0, 102.0, 103.0, 100.0, 100.0, 100.0]]]).reshape(1, 2, 10)
    coords = np.ascontiguousarray(coords)
    flags = np.array([[1, 1, 1, 1, 1, 2, 2, 2, 2, 2]], np.uint8)
    c = torch.from_numpy(coords).cuda()
    out, pruned = pea.dense_pointset._launch_prune(c, flags, np.array([[[10, 10], [20, 10], [20, 20], [10, 20]]], np.int32),
                                                   np.array([[[50, 100, 60, 100]]], np.int32), np.array([1], np.int32))
    # the last point: x = -1 wraps to column 223, far from the band
    assert pruned.cpu().numpy().tolist() == [[1, 1, 1, 0, 0, 1, 0, 1, 0, 0]]
    assert out is c and np.array_equal(c.cpu().numpy()[0, :, 3], [20.000001, 15.0]) and not c[0, :, 0].any()


def test_timing_hooks(pea):
    import ctypes
    import torch
    from pose_estimation_amd import _abi
    q = torch.randn(2, 32, 256, device="cuda")
    _abi.check(_abi.lib.ispa_timing_enable(3))
    for _ in range(4):                                   # the 4th call finds no free slot
        pea.correspondence.dense_nn(q, q, 1)
    ms = (ctypes.c_float * 8)()
    n = ctypes.c_int(0)
    _abi.check(_abi.lib.ispa_timing_collect(ms, 8, ctypes.byref(n)))
    assert n.value == 3 and all(0.0 < ms[i] < 100.0 for i in range(3))
    _abi.check(_abi.lib.ispa_timing_enable(0))
    _abi.check(_abi.lib.ispa_timing_collect(ms, 8, ctypes.byref(n)))
    assert n.value == 0


def test_errors_are_loud(pea):
    import torch
    from pose_estimation_amd import _abi
    # a batch of poses against ONE depth / key image would index past the image: refused in Python
    pr3 = np.zeros((3, 3, 10))
    with pytest.raises(ValueError):
        pea.zbuffer.visibility(pr3, np.zeros((540, 960), np.float32))
    with pytest.raises(ValueError):
        pea.zbuffer.visibility(pr3, np.zeros((2, 540, 960), np.int64))
    assert pea.zbuffer.visibility(pr3, np.zeros((3, 54, 96), np.float32)).shape == (3, 10)
    with pytest.raises(_abi.IspaError):
        _abi.check(_abi.lib.ispa_nn_topk(None, None, 0, 1, 8, 16, 16, 1, 1, None, None, None, 0, None, None))
    q = torch.zeros(1, 8, 16, device="cuda")
    ws = torch.empty(256, dtype=torch.uint8, device="cuda")
    rc = _abi.lib.ispa_nn_topk(q.data_ptr(), q.data_ptr(), 0, 1, 8, 16, 16, 1, 1, q.data_ptr(), q.data_ptr(),
                               ws.data_ptr(), 16, None, None)
    assert rc == -3 and b"workspace" in _abi.lib.ispa_last_error()
