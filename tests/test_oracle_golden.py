"""CPU: the oracle restatement against golden vectors produced by the imported reference."""
import hashlib

import numpy as np
import pytest

from oracle import cnative, correlation, disparity, geometry as g, keypoints as kp, nn, prune2d, zbuffer

ANCHORS = ["bed", "chair", "sofa_b", "sofa", "table"]


def test_camera_matrices(golden_geometry):
    z = golden_geometry
    for pose, e, p, r in zip(z["g1_poses"], z["g1_E"], z["g1_P"], z["g1_R"]):
        assert np.array_equal(g.rot_three_angles(pose[0], pose[1], pose[2]), r)
        assert np.array_equal(g.ex_camera_matrix(*pose), e)
        assert np.array_equal(g.camera_from_params(pose), p)


def test_survey_known_answer():
    e = g.ex_camera_matrix(30, 10, 0, 3)
    assert np.allclose(e[2], [-0.85286853, -0.49240388, -0.17364818, 3.0], atol=1e-8)
    p = g.camera_from_params([30, 10, 0, 3])
    pts = np.random.RandomState(0).uniform(-0.4, 0.4, (5, 3))
    r = g.get_render_location(pts, p)
    assert r[0].tolist() == [448., 439., 382., 489., 584.]
    assert r[1].tolist() == [210., 289., 155., 185., 155.]
    t, rc = g.crop_visibility_array(r.copy(), [100, 400, 300, 700])
    assert t.all() and rc[0].tolist() == [83., 78., 46., 106., 159.]
    assert rc[1].tolist() == [90., 134., 59., 76., 59.]


@pytest.mark.parametrize("name", ANCHORS)
def test_projection_anchors(golden_geometry, name):
    z = golden_geometry
    cam = z["g2_big_cam"]
    want = z[f"g2_anchor_{name}_render"]
    assert np.array_equal(g.get_render_location(z[f"g2_anchor_{name}"], cam), want)
    assert np.array_equal(cnative.project_fma(z[f"g2_anchor_{name}"], cam), want)


def test_projection_mesh_and_cloud(golden_geometry):
    z = golden_geometry
    cam = z["g2_big_cam"]
    assert np.array_equal(g.get_render_location(z["g2_obj_vertices"], cam), z["g2_obj_render"])
    got_c = cnative.project_fma(z["g2_obj_vertices"], cam)
    assert np.array_equal(got_c[:2], z["g2_obj_render"][:2])
    assert np.allclose(got_c[2], z["g2_obj_render"][2], rtol=1e-14, atol=0)
    big = np.random.RandomState(0).uniform(-0.4, 0.4, (100000, 3))
    for fn in (g.get_render_location, cnative.project_fma):
        r = fn(big, cam)
        sub = z["g2_big_sub_index"]
        assert np.array_equal(r[:2, sub], z["g2_big_sub_render"][:2])
        assert np.allclose(r[2, sub], z["g2_big_sub_render"][2], rtol=1e-14, atol=0)
        digest = hashlib.sha256(np.ascontiguousarray(r[:2]).tobytes()).digest()
        assert np.array_equal(np.frombuffer(digest, dtype=np.uint8), z["g2_big_uv_sha256"])


@pytest.mark.parametrize("tag", ["near", "std", "tilt"])
@pytest.mark.parametrize("dname", ["surface", "const", "plane"])
def test_depth_visibility(golden_geometry, tag, dname):
    z = golden_geometry
    r = z[f"g3_{tag}_render"]
    d = (np.full((540, 960), 3.0, np.float32) if dname == "const" else z[f"g3_{tag}_depth_{dname}"])
    ta, same = g.depth_visibility_array_a(r.copy(), d)
    assert np.array_equal(ta, z[f"g3_{tag}_{dname}_truth_a"])
    tb, pruned = g.depth_visibility_array_b(r.copy(), d)
    assert np.array_equal(tb, z[f"g3_{tag}_{dname}_truth_b"])
    assert np.array_equal(pruned, z[f"g3_{tag}_{dname}_pruned_b"])
    # the fixture must exercise both outcomes and the frame border
    if dname == "surface":
        assert 0 < ta.sum() < ta.size and 0 < tb.sum() < tb.size


def test_depth_visibility_literal_matches_vectorised(golden_geometry):
    z = golden_geometry
    r = z["g3_near_render"][:, :800]
    d = z["g3_near_depth_surface"]
    assert np.array_equal(g.depth_visibility_literal(r, d, "A"), g.depth_visibility_array_a(r, d)[0])
    assert np.array_equal(g.depth_visibility_literal(r, d, "B"), g.depth_visibility_array_b(r, d)[0])


def test_depth_visibility_edges(golden_geometry):
    z = golden_geometry
    r, d = z["g3_edge_render"], z["g3_edge_depth"]
    assert np.array_equal(g.depth_visibility_array_a(r.copy(), d)[0], z["g3_edge_truth_a"])
    assert np.array_equal(g.depth_visibility_array_b(r.copy(), d)[0], z["g3_edge_truth_b"])
    # x = 950 and y = 0 / 540 are outside the gate; x = 949, y = 539 inside
    assert not z["g3_edge_truth_a"][7] and z["g3_edge_truth_a"][6]


@pytest.mark.parametrize("tag", ["wide", "tall", "square", "odd"])
def test_crop_remap(golden_geometry, tag):
    z = golden_geometry
    arr = z["g3_std_render"].copy()
    t, same = g.crop_visibility_array(arr, z[f"g4_{tag}_crop"].tolist())
    assert same is arr                      # in-place contract
    assert np.array_equal(t, z[f"g4_{tag}_truth"])
    assert np.array_equal(arr, z[f"g4_{tag}_render"])


def test_crop_noresize(golden_geometry):
    z = golden_geometry
    arr = z["g3_std_render"].copy()
    t, _ = g.crop_visibility_array(arr, [100, 400, 300, 700], resize=False)
    assert np.array_equal(t, z["g4_noresize_truth"]) and np.array_equal(arr, z["g4_noresize_render"])


@pytest.mark.parametrize("tag", ["cloud", "mesh", "edge"])
def test_make_disparity(golden_disparity, tag):
    z = golden_disparity
    args = (z[f"{tag}_pts"], z[f"{tag}_params"].tolist(), z[f"{tag}_depth"], z[f"{tag}_crop"].tolist())
    want = z[f"{tag}_flow"]
    assert np.array_equal(disparity.make_disparity_core(*args, literal=True), want)
    assert np.array_equal(disparity.make_disparity_core(*args), want)
    assert np.count_nonzero(want) > 1000


def test_splat_python_slice_semantics():
    # negative starts wrap; a start that wraps past the stop gives an empty slice
    uv = np.array([[0, 1, 2, 223, 230, -5, 100], [50, 50, 50, 223, 10, 60, -3]], dtype=np.float64)
    delta = np.arange(14, dtype=np.float64).reshape(2, 7) + 1
    a = disparity.splat_flow_literal(uv, delta)
    b = disparity.splat_flow(uv, delta)
    assert np.array_equal(a, b)
    assert a[50, 0, 0] == 3.0 and a[48:52, 0:4, 0].tolist() == [[3.0] * 4] * 4   # only u=2 writes
    assert a[221:224, 221:224, 0].min() == 4.0
    assert a[58:62, 217:221, 0].min() == 6.0                                     # u=-5 wraps
    assert g.py_slice_bounds(1, 2, 224) == (223, 223) and g.py_slice_bounds(-5, 2, 224) == (217, 221)


@pytest.mark.parametrize("tag", ["cloud", "mesh", "border", "shipped"])
def test_demo_flow(golden_demo_flow, tag):
    """The flow demo (render_wt_pt_proj/get_of.py:38-62) re-typed on the imported reference primitives: 540 x 960
    maps, no crop, points visible at both poses; 'shipped' runs against the depth render the reference ships."""
    from conftest import dense_flow
    z = golden_demo_flow
    want = dense_flow(z, tag)
    args = (z[f"{tag}_pts"], z[f"{tag}_params"].tolist(), z[f"{tag}_depth"])
    assert np.array_equal(disparity.demo_flow_core(*args, literal=True), want)
    assert np.array_equal(disparity.demo_flow_core(*args), want)
    assert np.count_nonzero(want) > 1000


@pytest.mark.parametrize("name", ANCHORS)
def test_get_render_array_variant_b(golden_disparity, golden_geometry, name):
    z = golden_disparity
    fd = {"azimuth": 120.0, "elevation": 10.0, "tilt": 0.0, "rho": 2.2, "crop_params": [60, 480, 250, 720]}
    truth, crop_pts = g.get_render_array(fd, golden_geometry[f"g2_anchor_{name}"], z[f"rb_{name}_depth"])
    assert np.array_equal(truth, z[f"rb_{name}_truth"])
    assert np.array_equal(crop_pts, z[f"rb_{name}_crop_pts"])


def test_pt_depth_map(golden_disparity):
    z = golden_disparity
    assert np.array_equal(g.get_pt_depth_map(z["ptdepth_render"]), z["ptdepth_map"])


@pytest.mark.parametrize("tag", ["b1k1", "b2k10", "b3k14"])
def test_correlation_eccv(golden_correlation, tag):
    z = golden_correlation
    f1 = z[f"{tag}_f1"].astype(np.float32)
    f2 = z[f"{tag}_f2"].astype(np.float32)
    got = correlation.correlation_map_eccv(f1, f2, z[f"{tag}_knn"])
    want = z[f"{tag}_out"]
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.allclose(got, want, rtol=1e-5, atol=1e-8)


def test_bilinear_resize(golden_correlation):
    z = golden_correlation
    got = correlation.bilinear_align_corners(z["resize_in"], 28, 28)
    # last-bit differences only (torch contracts multiply-adds in its own order)
    assert np.allclose(got, z["resize_out"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["b1c128", "b2c32"])
def test_correlation_acmm_golden(golden_correlation_acmm, tag):
    """net_final_old.py:33-61 run unmodified on CPU (stub self) pins the restatement, NaN column included."""
    z = golden_correlation_acmm
    f1 = z[f"{tag}_f1"].astype(np.float32)
    f2 = z[f"{tag}_f2"].astype(np.float32)
    got = correlation.correlation_map_acmm(f1, f2, z[f"{tag}_mask"].astype(np.float32))
    want = z[f"{tag}_out"]
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want[-1, :, 0, 0]).all()
    # a correlation is a sum of 128 signed products of unit-vector components: entries near the relu
    # threshold are cancellation residues, whose float32 summation-order error (1e-7) is absolute
    assert np.allclose(got, want, rtol=1e-5, atol=5e-7, equal_nan=True)


def test_tables_equal_the_imported_reference(golden_tables):
    """skeleton_map / leg_range / syn_template / img_size as data_creator/global_variables.py holds them
    (written by tests/golden/make_golden.py from the imported module) against both typed copies."""
    import ast
    import os
    t = golden_tables
    assert set(kp.SKELETON_MAP) == set(t["skeleton_map"]) and "car" in t["skeleton_map"]
    for cls, (bones, tags, seat) in t["skeleton_map"].items():
        o = kp.SKELETON_MAP[cls]
        assert [list(b) for b in o[0]] == bones and list(o[1]) == tags and list(o[2]) == seat, cls
    assert {k: list(v) for k, v in prune2d.LEG_RANGE.items()} == t["leg_range"]
    # the product's copies: read as literals from the source files (importing the package needs the HIP library)
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pose_estimation_amd")

    def module_values(path, names):
        env = {}
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.Assign) and any(getattr(tg, "id", None) in names or
                                                    (isinstance(tg, ast.Subscript) and getattr(tg.value, "id", None) in names)
                                                    for tg in node.targets):
                exec(compile(ast.Module([node], []), path, "exec"), env)
        return env
    env = module_values(os.path.join(root, "dense_pointset.py"), {"skeleton_map", "leg_range", "img_size", "skeleton_size"})
    assert set(env["skeleton_map"]) == set(t["skeleton_map"])
    for cls, (bones, tags, seat) in t["skeleton_map"].items():
        o = env["skeleton_map"][cls]
        assert [list(b) for b in o[0]] == bones and list(o[1]) == tags and list(o[2]) == seat, cls
    assert {k: list(v) for k, v in env["leg_range"].items()} == t["leg_range"]
    assert list(env["img_size"]) == t["img_size"] and env["skeleton_size"] == t["skeleton_size"]
    env = module_values(os.path.join(root, "global_variables.py"), {"syn_template"})
    assert {k: list(v) for k, v in env["syn_template"].items()} == t["syn_template"]


def test_acmm_shares_contraction_with_eccv(golden_correlation):
    z = golden_correlation
    f1 = z["b2k10_f1"].astype(np.float32)
    f2 = z["b2k10_f2"].astype(np.float32)
    mask = np.ones((2, 1, 56, 56), np.float32)
    acmm = correlation.correlation_map_acmm(f1, f2, mask)
    # rows p of acmm(f1, f2) un-normalised == eccv(real=f2, template=f1) at template cell p
    knn = np.array([[3, 5], [27, 0], [0, 27]])
    f1r = correlation.bilinear_align_corners(f1, 28, 28)
    f2r = correlation.bilinear_align_corners(f2, 28, 28)
    raw = np.maximum(np.einsum("bcp,bcyx->bpyx", f1r.reshape(2, -1, 784), f2r), 0)
    norm = np.sqrt((raw ** 2).sum(1, keepdims=True))
    for x, y in knn:
        p = y * 28 + x
        assert np.allclose(acmm[:, p], (raw / norm)[:, p], rtol=1e-5, atol=1e-7)
    assert np.allclose(np.sqrt((acmm ** 2).sum(1)), 1.0, atol=1e-4)


def test_l2norm():
    x = np.random.RandomState(3).randn(2, 16, 5, 5).astype(np.float32)
    y = correlation.l2norm(x)
    assert np.allclose((y ** 2).sum(1), 1.0, atol=1e-6)


def test_nn_exact_c_equals_numpy():
    rs = np.random.RandomState(5)
    q = rs.randn(24, 150).astype(np.float32)
    r = rs.randn(24, 333).astype(np.float32)
    r[:, 40] = r[:, 7]                       # exact duplicate -> tie, lowest index must win
    q[:, 3] = r[:, 7]
    for k in (1, 5):
        i_np, d_np = nn.nn_exact_numpy(q, r, k)
        i_c, d_c = cnative.nn_exact(q, r, k)
        assert np.array_equal(i_np, i_c) and np.array_equal(d_np, d_c)
    i1, d1 = cnative.nn_exact(q, r, 2)
    assert i1[0, 3] == 8 and i1[1, 3] == 41 and d1[0, 3] == 0.0   # 1-based
    i0, _ = cnative.nn_exact(q, r, 1, index_base=0)
    assert np.array_equal(i0 + 1, i1[:1])


def test_round_bf16():
    x = np.array([1.0, 1.00390625, 1.005859375, -3.1415927, 1e-40, np.inf], np.float32)
    y = cnative.round_bf16(x)
    assert y[0] == 1.0 and y[1] == 1.0 and y[2] == 1.0078125 and np.isinf(y[5])
    assert (y.view(np.uint32) & 0xFFFF == 0).all()


def test_zbuffer_oracle():
    pr = np.array([[10, 10, 10, 500, -1, 959], [20, 20, 20, 100, 5, 539], [2.0, 1.5, 1.5, 3.0, 1.0, 4.0]])
    key, win, depth = zbuffer.zbuffer(pr, 540, 960)
    assert win[20, 10] == 1 and depth[20, 10] == np.float32(1.5)   # lowest index among equal depth
    assert win[100, 500] == 3 and win[539, 959] == 5 and win[0, 0] == -1
    vis = zbuffer.zbuffer_visibility(pr, depth)
    assert vis.tolist() == [False, True, True, True, False, True]


def test_filename_codec_and_keypoints(golden_geometry):
    name = "04256520_1fd45c57ab27cb6cea65c47b660136e7_a90@179_e0@236_t0@000_d3@000_199_343_385_575_.jpg"
    d = kp.get_params_from_syn_file("x/y/" + name)
    assert d["azimuth"] == 90.179 and d["tilt"] == -0.0 and d["crop_params"] == [199, 343, 385, 575]
    assert kp.img_to_depth_file(name) == name[:-len("_199_343_385_575_.jpg")] + ".png1.exr"
    assert kp.render_file_name(d) == name.replace("a90@179", "a90@179").replace("_t0@000", "_t-0@000") \
        or kp.render_file_name(d) == name
    chair = golden_geometry["g2_anchor_chair"]
    dense = kp.dense_skeleton_3d(chair, "chair")
    assert dense.shape == (3, 130)
    assert np.array_equal(dense[:, 0], chair[4]) and np.array_equal(dense[:, 9], chair[0])
    grid = kp.knn_inds_to_grid(np.array([[-4.0, 10.0], [223.9, 300.0], [8.0, 15.99]]))
    assert grid.tolist() == [[0, 1], [27, 27], [1, 1]]
    xy = kp.ind2coord(np.array([[[1, 57, 3136]]]), 56)
    assert xy[0, 0].tolist() == [[0, 0], [0, 4], [220, 220]]
    params = np.array([[0, 10, 0], [350, 0, 0], [120, 10, 0], [10, 80, 0]], dtype=np.float64)
    assert kp.nearest_view([355, 0, 0], params) == 1 and kp.nearest_view([2, 9, 0], params) == 0


def test_extract_feature_known_answers():
    from oracle import nn as onn
    c, h, w = 3, 6, 7
    f = (100.0 * np.arange(c)[:, None, None] + 10.0 * np.arange(h)[None, :, None] + np.arange(w)[None, None, :])
    f = f[None].astype(np.float32)
    coords = np.array([[[8.0, 12.0, 0, 0, 1], [10.0, 14.0, 0, 0, 1], [-3.0, 100.0, 0, 0, 1]]], dtype=np.float32)
    near = onn.extract_feature(f, coords)
    assert near.shape == (1, 3, 3)
    assert np.array_equal(near[0, :, 0], [32.0, 132.0, 232.0])          # cell (x=2, y=3)
    assert np.array_equal(near[0, :, 1], [32.0, 132.0, 232.0])          # floor(2.5), floor(3.5)
    assert np.array_equal(near[0, :, 2], [50.0, 150.0, 250.0])          # clamped to (x=0, y=5)
    bil = onn.extract_feature(f, coords, mode="bilinear")
    assert np.array_equal(bil[0, :, 0], [32.0, 132.0, 232.0])
    assert np.array_equal(bil[0, :, 1], [37.5, 137.5, 237.5])           # a linear ramp is reproduced exactly


def test_prune2d_known_answers():
    """Hand-checked cases of the keypoint pruning restatement (parity unpinned: the reference decides
    these through cv2 rasterisers; see oracle/prune2d.py)."""
    square = np.array([[10, 10], [20, 10], [20, 20], [10, 20]], np.int32)
    assert prune2d.point_in_polygon(15.0, 15.0, square)
    assert prune2d.point_in_polygon(10.0, 12.5, square)            # on an edge
    assert prune2d.point_in_polygon(20.0, 20.0, square)            # on a corner
    assert not prune2d.point_in_polygon(20.000001, 15.0, square)
    assert not prune2d.point_in_polygon(5.0, 10.0, square)         # level with an edge, outside
    assert not prune2d.point_in_polygon(0.0, 0.0, square)
    bowtie = np.array([[0, 0], [10, 10], [10, 0], [0, 10]], np.int32)
    assert prune2d.point_in_polygon(2.0, 5.0, bowtie) and prune2d.point_in_polygon(8.0, 5.0, bowtie)
    assert not prune2d.point_in_polygon(5.0, 2.0, bowtie)
    # a horizontal 5-pixel line paints 5 rows; its caps reach 2 pixels past the ends on the axis
    src = np.zeros((224, 224), np.uint8)
    prune2d.draw_band(src, (50, 100), (60, 100), 5)
    assert np.array_equal(np.nonzero(src[:, 55])[0], np.arange(98, 103))
    assert src[100, 48] == 255 and src[100, 47] == 0 and src[100, 62] == 255 and src[100, 63] == 0
    assert src[98, 48] == 0 and src[99, 48] == 255 and src[98, 49] == 255    # cap: 1 + 4 <= 6.25 < 4 + 4
    # 45 degrees: pixels at perpendicular distance 2.12 are in, 2.83 out
    src[:] = 0
    prune2d.draw_band(src, (10, 10), (40, 40), 5)
    assert src[20, 23] == 255 and src[23, 20] == 255 and src[19, 23] == 0
    # a chair facing right: left-side bones are drawn, right and centre bones are tested
    rs = np.random.RandomState(0)
    sparse = np.concatenate([rs.randint(0, 224, (10, 2)), np.ones((10, 1), int)], 1)
    out = prune2d.make_dense_keypoints_image(sparse, "chair")
    assert out.shape == (2, 130)
    assert prune2d.make_dense_keypoints_image(np.zeros((10, 3)), "chair", dataset="pascal3D") is None


def test_exr_reader_on_the_blender_render_and_round_trips(tmp_path):
    """pose_estimation_amd.exr against (i) a 48-line crop of the one depth render the reference ships (Blender,
    ZIP, FLOAT B G R; compressed blocks byte for byte, see tests/golden/make_golden.py), whose decoded values
    were recorded from the whole file, (ii) the whole file when the reference is present, (iii) its own writer
    for every supported compression, HALF and UINT channels, odd sizes."""
    import hashlib
    import json
    import os
    from pose_estimation_amd import exr, run_proj
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    info = json.load(open(os.path.join(here, "exr_reference_file.json")))
    crop = exr.read_exr(os.path.join(here, "blender_depth_crop.exr"))
    assert sorted(crop) == info["channels"] and crop["R"].shape == (48, 960) and crop["R"].dtype == np.float32
    assert hashlib.sha256(crop["R"].tobytes()).hexdigest() == info["sha256_R_rows_256_303"]
    assert np.array_equal(crop["R"], crop["G"]) and float(crop["R"].max()) == info["max"] == 10.0
    assert float(crop["R"].min()) == info["crop_min"] and int((crop["R"] < 10.0).sum()) == info["crop_object_pixels"]
    for y, x, v in info["probes"]:
        if 256 <= y < 304:
            assert float(crop["R"][y - 256, x]) == v
    full = "/root/reference/data_creator/render_wt_pt_proj/z_map/temp0001.exr"
    if os.path.exists(full):
        r = run_proj.get_depthmap_from_exr(full)                  # the reference's entry point, run_proj.py:49-57
        assert r.shape == tuple(info["shape"]) and r.dtype == np.float32
        assert hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest() == info["sha256_R"]
        assert all(float(r[y, x]) == v for y, x, v in info["probes"])
    rs = np.random.RandomState(3)
    for comp in ("NONE", "RLE", "ZIPS", "ZIP"):
        for shape in ((37, 53), (16, 8), (1, 1)):
            depth = rs.uniform(0.5, 20.0, shape).astype(np.float32)
            depth[: shape[0] // 2] = 20.0                          # runs of equal bytes for RLE
            ids = rs.randint(0, 2 ** 31, shape).astype(np.uint32)
            path = str(tmp_path / f"t_{comp}_{shape[0]}.exr")
            exr.write_exr(path, {"R": depth, "Z.id": ids}, compression=comp)
            back = exr.read_exr(path)
            assert np.array_equal(back["R"], depth) and np.array_equal(back["Z.id"], ids)
            assert np.array_equal(run_proj.get_depthmap_from_exr(path), depth)
            exr.write_exr(path, {"R": depth}, compression=comp, half=True)
            assert np.array_equal(exr.read_exr_channel(path), depth.astype(np.float16).astype(np.float32))
    exr.write_exr(str(tmp_path / "layer.exr"), {"Image.R": np.ones((4, 4), np.float32)})
    assert exr.read_exr_channel(str(tmp_path / "layer.exr"), "R").shape == (4, 4)
    with pytest.raises(KeyError):
        exr.read_exr_channel(str(tmp_path / "layer.exr"), "Z")
    with pytest.raises(ValueError):
        exr.read_exr(b"not an exr file at all")


def test_hard_negative_rule_by_hand():
    """hard_negative_processing_layer.py:188-192,246-250 on a 4-cell-wide map: 1-based indices 6, 1, 16, 11 are
    the cells (1,1), (0,0), (3,3), (2,2); ground truth in image pixels (stride 4); threshold 4 feature cells."""
    ind = np.array([[[6, 1, 16, 11]]])
    corr = np.array([[[0, 0, 4.0, 4.0, 1],        # on its cell: distance 0
                      [0, 0, 0.0, 16.0, 1],       # (0,0) against (0,4): exactly 4, not a hard negative
                      [0, 0, 0.0, 0.0, 1],        # (3,3) against (0,0): sqrt(18) > 4
                      [0, 0, 40.0, 40.0, 0]]])    # far off, but not a positive pair (flag 0)
    mask, xy = kp.hard_negatives(ind, corr, 4)
    assert mask.tolist() == [[False, False, True, False]]
    assert np.array_equal(xy[0], [[4, 4], [0, 0], [12, 12], [8, 8]])
    assert kp.hard_negatives(ind, corr, 4, thresh=3.9)[0].tolist() == [[False, True, True, False]]


@pytest.mark.parametrize("tag", ["b1c128", "b2c64"])
@pytest.mark.parametrize("k", [1, 5])
def test_nn_oracle_against_the_reference_contraction(golden_nn_crosspin, tag, k):
    """The dense nearest-neighbour contract against numbers the reference itself produced: its all-pairs
    contraction (eccv_net.py:193-210, run unbound on CPU by tests/golden/make_golden.py) ranks the cells of
    feature1 for every template cell by dot product, which on unit-norm features is the ranking by L2 distance.
    Both restatements of the contract (numpy and C) must return the reference's best cell (k = 1) and its five
    best cells in order (k = 5) wherever the reference's own float32 values decide the ranking."""
    from conftest import crosspin_cases
    from oracle import cnative, nn as onn
    checked = 0
    for q, r, want, decided in crosspin_cases(golden_nn_crosspin, tag, k):
        ind_np, dist_np = onn.nn_exact_numpy(q, r, k, 1)
        ind_c, dist_c = cnative.nn_exact(q, r, k, 1)
        assert np.array_equal(ind_np, ind_c) and np.array_equal(dist_np, dist_c)
        assert np.array_equal(ind_c[:, decided], want[:, decided])
        assert decided.mean() > (0.97 if k == 1 else 0.85)
        checked += int(decided.sum())
    assert checked > 600


def test_zbuffer_oracle_against_the_blender_render():
    """The z-buffer contract against a physical measurement the reference ships: Blender's depth render of the bed
    mesh (z_map/temp0001.exr) against the z-buffer of 800k surface samples of that mesh at the fitted pose
    (azimuth 330, elevation 25, distance 2).  Approximate by nature - nearest sample per pixel against a ray
    through the pixel centre: the object masks must overlap to 97 %, the depths agree to a median of 0.003 and to
    0.01 on 88 % of the common pixels (measured: 98.3 %, 0.0012, 92 %)."""
    from conftest import physical_zbuffer_agreement, physical_zbuffer_case
    from oracle import geometry as og, zbuffer as ozb
    pts, pose, shipped, stats = physical_zbuffer_case()
    depth = ozb.zbuffer(og.get_3d_to_2d(pts, pose), 540, 960, radius=0)[2]
    iou, med, near = physical_zbuffer_agreement(depth, shipped)
    assert iou > 0.97 and med < 0.003 and near > 0.88, (iou, med, near)
    assert abs(iou - stats["iou"]) < 1e-3 and abs(near - stats["within_0.01"]) < 1e-3
