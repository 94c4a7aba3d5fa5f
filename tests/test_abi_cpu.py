"""CPU: the C-ABI library loads, exports every symbol the header declares, and rejects bad calls
before touching a device."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(HERE, "include", "ispa_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(ispa_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound():
    from pose_estimation_amd import _abi
    names = _declared()
    assert names == set(_abi.SIGNATURES), names ^ set(_abi.SIGNATURES)
    lib = ctypes.CDLL(_abi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert _abi.lib.ispa_abi_version() == 1


def test_argument_checks_do_not_need_a_gpu():
    from pose_estimation_amd import _abi
    assert _abi.lib.ispa_nn_workspace_bytes(1, 128, 4096, 4096, 1) > 0
    assert _abi.lib.ispa_nn_workspace_bytes(1, 128, 4096, 4096, 9) == 0
    rc = _abi.lib.ispa_nn_topk(None, None, 0, 1, 8, 16, 16, 9, 1, None, None, None, 0, None, None)
    assert rc == -1 and b"k = 9" in _abi.lib.ispa_last_error()
    rc = _abi.lib.ispa_depth_visibility(1, 10, 1, 1, 0, 100, 100, 540, 950, 0.15, 0, 1, None)
    assert rc == -1 and b"gate" in _abi.lib.ispa_last_error()
    rc = _abi.lib.ispa_project(None, 0, None, 0, None, None)
    assert rc == 0                                           # empty input is a no-op
    rc = _abi.lib.ispa_nearest_view(1, 1, 1, 0, 1, None, None)
    assert rc == -1 and b"no stored poses" in _abi.lib.ispa_last_error()
    rc = _abi.lib.ispa_pck_counts(1, 1, 1, 4, 1, 56, 4.0, 1, 1, 9, 1, None)
    assert rc == -1 and b"radii" in _abi.lib.ispa_last_error()
    assert _abi.lib.ispa_pck_counts(None, 0, 1, 0, None, 56, 4.0, 1, None, 0, None, None) == 0
    rc = _abi.lib.ispa_extract_feature(1, 1, 8, 4, 4, 1, 10, 5, 4, 4.0, 0, 1, None)
    assert rc == -1 and b"coordinate columns" in _abi.lib.ispa_last_error()
    rc = _abi.lib.ispa_prune_points_2d(1, 1, 10, 1, 1, 65, None, None, 0, 5, 224, 224, None, None)
    assert rc == -1 and b"vertices" in _abi.lib.ispa_last_error()
    rc = _abi.lib.ispa_prune_points_2d(1, 1, 10, 1, 1, 4, None, None, 0, 5, 224, 9000, None, None)
    assert rc == -1 and b"image size" in _abi.lib.ispa_last_error()
    assert _abi.lib.ispa_prune_points_2d(None, 0, 0, None, None, 4, None, None, 0, 5, 224, 224, None, None) == 0


def test_host_helpers_without_gpu():
    import numpy as np
    from pose_estimation_amd import correspondence, dense_pointset, io_utils, pt2render
    from oracle import geometry as og, keypoints as okp
    assert np.array_equal(pt2render.ex_camera_matrix(30, 10, 0, 3), og.ex_camera_matrix(30, 10, 0, 3))
    poses = np.array([[30, 10, 0, 3], [200, -5, 12, 2.5]], dtype=float)
    want = np.stack([og.camera_from_params(p) for p in poses])
    assert np.allclose(pt2render.camera_matrices(poses), want, rtol=1e-13, atol=1e-12)
    name = "04256520_1fd45c57ab27cb6cea65c47b660136e7_a90@179_e0@236_t5@000_d3@000_199_343_385_575_.jpg"
    d = io_utils.get_params_from_file("a/b/" + name)
    assert d == okp.get_params_from_syn_file(name) and d["tilt"] == -5.0
    assert io_utils.img_to_depth_file(name) == okp.img_to_depth_file(name)
    assert os.path.basename(io_utils.get_final_file_name(d)) == name == okp.render_file_name(d)
    assert io_utils.get_depth_map_name(d) == okp.depth_map_name(d)
    rs = np.random.RandomState(0)
    for cls, k in (("chair", 10), ("bed", 10), ("sofa", 14), ("table", 8), ("swivelchair", 13)):
        a = rs.randn(k, 3)
        coords = dense_pointset.get_additional_coords_keypoint_5(a.T, cls)
        assert np.array_equal(dense_pointset.dense_skeleton(coords, cls), okp.dense_skeleton_3d(a, cls))
        assert np.array_equal(dense_pointset.dense_skeleton(coords[:2], cls, dim=2), okp.dense_skeleton_2d(coords[:2], cls))
    xy = rs.uniform(-20, 260, (14, 2))
    assert np.array_equal(correspondence.knn_inds_to_grid(xy), okp.knn_inds_to_grid(xy))
    ind = rs.randint(1, 3137, (2, 1, 50))
    assert np.array_equal(correspondence.ind2coord(ind, 56), okp.ind2coord(ind, 56))


def test_spatial_order_host_logic():
    import numpy as np
    from pose_estimation_amd import run_proj
    rs = np.random.RandomState(5)
    pts = rs.uniform(-1, 1, (5000, 3))
    step = lambda p: np.linalg.norm(np.diff(p, axis=0), axis=1).mean()
    for curve in ("hilbert", "morton"):
        perm = run_proj.spatial_order(pts, curve=curve)
        assert perm.dtype == np.int64 and sorted(perm.tolist()) == list(range(5000))
        assert step(pts[perm]) < 0.25 * step(pts)            # neighbours in the array are neighbours in space
    # the Hilbert curve visits a full grid in unit steps (no jumps, unlike the Z-curve)
    grid = np.stack(np.meshgrid(*[np.arange(8)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(float)
    walk = grid[run_proj.spatial_order(grid, bits=3, curve="hilbert")]
    assert (np.abs(np.diff(walk, axis=0)).sum(1) == 1).all()
    perm = run_proj.spatial_order(pts)
    assert np.array_equal(run_proj.spatial_order(pts), perm)  # deterministic
    flat = pts.copy()
    flat[:, 2] = 0.5                                          # a degenerate axis must not divide by zero
    assert sorted(run_proj.spatial_order(flat).tolist()) == list(range(5000))


def test_surface_sampler_without_trimesh(tmp_path):
    """run_proj.get_keypoints: area-weighted surface samples of an .obj (a unit cube with one face as a quad and
    negative indices) lie on the surface and cover the faces in proportion to their area."""
    import numpy as np
    from pose_estimation_amd import run_proj
    obj = tmp_path / "cube.obj"
    obj.write_text("\n".join(
        ["v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0 0 2", "v 1 0 2", "v 1 1 2", "v 0 1 2",
         "f 1 2 3 4", "f 5/1/1 6/2/1 7/3/1 8/4/1", "f 1 2 6 5", "f 2 3 7 6", "f 3 4 8 7", "f -5 -8 -4 -1"]))
    verts, faces = run_proj.load_obj(str(obj))
    assert verts.shape == (8, 3) and faces.shape == (12, 3) and faces.min() == 0 and faces.max() == 7
    pts = run_proj.get_keypoints(str(obj), 40000, seed=1)
    assert pts.shape == (40000, 3) and np.array_equal(pts, run_proj.get_keypoints(str(obj), 40000, seed=1))
    on = (np.isclose(pts[:, 0], 0) | np.isclose(pts[:, 0], 1) | np.isclose(pts[:, 1], 0) | np.isclose(pts[:, 1], 1)
          | np.isclose(pts[:, 2], 0) | np.isclose(pts[:, 2], 2))
    assert on.all() and pts.min() >= -1e-12 and pts[:, 2].max() <= 2 + 1e-12
    # areas: bottom / top 1 each, four sides 2 each -> a tenth of the samples on the bottom face
    assert abs(np.isclose(pts[:, 2], 0).mean() - 0.1) < 0.01
    # box_mesh / chair_mesh (the synthetic model of the geometry benches): closed boxes inside the reference's cube
    verts, faces = run_proj.box_mesh([(0, 0, 0, 1, 2, 3)])
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    assert faces.shape == (12, 3) and np.isclose(0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum(), 22.0)
    cv, cf = run_proj.chair_mesh()
    assert cv.shape[0] == 8 * cf.shape[0] // 12 and np.abs(cv).max() <= 0.4
    chair = run_proj.sample_surface(cv, cf, 5000, seed=3)
    assert chair.shape == (5000, 3) and np.abs(chair).max() <= 0.4 + 1e-12


def test_one_hip_runtime_whatever_the_import_order():
    """The library and torch must end up on the same libamdhip64 (they exchange device pointers and streams): the
    package imports torch before it loads libispa_hip.so, so importing the package first must not bring in the
    system copy of the runtime beside torch's."""
    import subprocess
    import sys
    code = ("import pose_estimation_amd, re\n"
            "paths = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
            "print(len(paths), sorted(paths))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout


def test_cameras_from_params_matches_the_per_pose_functions_bit_for_bit():
    import numpy as np
    from pose_estimation_amd import run_proj
    rs = np.random.RandomState(21)
    poses = np.stack([rs.uniform(0, 360, 2000), rs.uniform(-30, 60, 2000), rs.uniform(-15, 15, 2000),
                      rs.uniform(1.5, 4, 2000)], 1)
    poses[:200, :3] = np.round(poses[:200, :3])
    want = np.stack([run_proj.camera_from_params(p) for p in poses])
    assert np.array_equal(run_proj.cameras_from_params(poses), want)
    assert run_proj.cameras_from_params(poses[0]).shape == (1, 3, 4)
