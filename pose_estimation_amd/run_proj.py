"""Projection orchestration with the reference's entry points.

Drop-in for data_creator/render_wt_pt_proj/run_proj.py:24-57 (get_3d_to_2d, prune_pts,
get_depthmap_from_exr, get_keypoints) plus batched forms for pose sweeps.
"""
import numpy as np
import torch

from . import _abi, _device as dv, exr, pt2render, variables
from .variables import VIS_XLIM, VIS_YLIM


def camera_from_params(params):
    ext = pt2render.ex_camera_matrix(params[0], params[1], params[2], params[3])
    return pt2render.get_camera_matrix(variables.INTRINSIC_CAMERA_MATRIX, ext, variables.WORLD_MATRIX)


_batched_cameras_exact = None


def cameras_from_params(params):
    """camera_from_params of every row of params (n,4) -> (n,3,4), bit for bit.

    pt2render.camera_matrices computes all rows at once (under 1 us per pose instead of 36); whether numpy's
    array routines round like its per-matrix ones is a property of the numpy build, so the first call compares
    the two on a fixed set of poses and the per-pose loop is kept if a single bit differs."""
    global _batched_cameras_exact
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    if _batched_cameras_exact is None:
        rs = np.random.RandomState(7)
        probe = np.stack([rs.uniform(0, 360, 64), rs.uniform(-30, 60, 64), rs.uniform(-15, 15, 64),
                          rs.uniform(1.5, 4, 64)], 1)
        probe[:16, :3] = np.round(probe[:16, :3])
        _batched_cameras_exact = np.array_equal(pt2render.camera_matrices(probe),
                                                np.stack([camera_from_params(p) for p in probe]))
    if _batched_cameras_exact:
        return pt2render.camera_matrices(params)
    return np.stack([camera_from_params(p) for p in params])


def load_obj(file_name):
    """Vertices (V,3) float64 and triangles (F,3) int64 of a Wavefront .obj (polygons are fanned; negative
    indices count from the end, as the format allows)."""
    verts, faces = [], []
    with open(file_name) as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(t.split("/")[0]) for t in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                faces += [[idx[0], idx[j], idx[j + 1]] for j in range(1, len(idx) - 1)]
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def get_keypoints(file_name, k=100000, seed=None):
    """run_proj.get_keypoints (run_proj.py:15-22): k points drawn uniformly from the surface of the mesh in
    file_name -> (k,3) float64.  The reference calls trimesh's mesh.sample(k), an area-weighted draw from its
    own random stream; this is the same distribution (triangle by area, uniform barycentric point) from
    numpy's, without trimesh.  The union of a multi-part mesh that the reference builds first is the set of
    all triangles, which is what an .obj read as one mesh already is."""
    verts, faces = load_obj(file_name)
    if len(faces) == 0:
        raise ValueError(f"{file_name} holds no faces")
    return sample_surface(verts, faces, k, seed)


def sample_surface(verts, faces, k=100000, seed=None):
    """k points drawn uniformly from the surface of a triangle mesh (verts (V,3), faces (F,3)): triangle by
    area, then a uniform barycentric point."""
    verts, faces = np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)
    rs = np.random.RandomState(seed)
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    tri = rs.choice(len(faces), size=int(k), p=area / area.sum())
    u, v = rs.uniform(size=(2, int(k)))
    flip = u + v > 1.0                                   # reflect into the triangle
    u[flip], v[flip] = 1.0 - u[flip], 1.0 - v[flip]
    return a[tri] + u[:, None] * (b[tri] - a[tri]) + v[:, None] * (c[tri] - a[tri])


def box_mesh(boxes):
    """Triangle mesh of axis-aligned boxes [(xmin, ymin, zmin, xmax, ymax, zmax), ...] -> verts (8B,3), faces (12B,3)."""
    corner = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], dtype=np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    verts, faces = [], []
    for n, bx in enumerate(np.asarray(boxes, dtype=np.float64)):
        verts.append(bx[:3] + corner * (bx[3:] - bx[:3]))
        for q in quads:
            faces += [[8 * n + q[0], 8 * n + q[1], 8 * n + q[2]], [8 * n + q[0], 8 * n + q[2], 8 * n + q[3]]]
    return np.concatenate(verts), np.asarray(faces, dtype=np.int64)


def chair_mesh():
    """A chair-sized stand-in for a CAD model (seat, back rest, four legs, two arm rests as boxes) inside the
    [-0.4, 0.4] cube the reference's normalised models occupy; for synthetic workloads."""
    legs = [(x, -0.40, z, x + 0.05, -0.05, z + 0.05) for x in (-0.30, 0.25) for z in (-0.30, 0.25)]
    return box_mesh([(-0.30, -0.05, -0.30, 0.30, 0.00, 0.30),            # seat
                     (-0.30, 0.00, -0.30, 0.30, 0.40, -0.25),           # back rest
                     (-0.30, 0.15, -0.25, -0.25, 0.18, 0.25),           # arm rests
                     (0.25, 0.15, -0.25, 0.30, 0.18, 0.25)] + legs)


def get_3d_to_2d(model_pts, params):
    """run_proj.py:24-31 -- params = [azimuth, elevation, tilt, distance] -> (3,N)."""
    return pt2render.get_render_location(model_pts, camera_from_params(params))


def get_depthmap_from_exr(file_name):
    """run_proj.py:49-57: the 'R' channel of a Blender depth render as an (H,W) float32 array.

    Decoded by pose_estimation_amd.exr (scanline OpenEXR, NONE / RLE / ZIPS / ZIP): neither OpenEXR nor
    Imath is needed.  A .npy file of the same image (file_name itself, or file_name + '.npy' next to a
    missing .exr) is read instead when given."""
    if file_name.endswith(".npy"):
        return np.load(file_name).astype(np.float32, copy=False)
    try:
        return exr.read_exr_channel(file_name, "R")
    except FileNotFoundError:
        try:
            return np.load(file_name + ".npy").astype(np.float32, copy=False)
        except FileNotFoundError:
            pass
        raise


def load_depth_stack(file_names, threads=8):
    """get_depthmap_from_exr of every file -> (n,H,W) float32, decoded on `threads` host threads (the inflate
    calls that dominate a decode run outside the interpreter lock)."""
    from concurrent.futures import ThreadPoolExecutor
    file_names = list(file_names)
    if threads <= 1 or len(file_names) <= 1:
        return np.stack([get_depthmap_from_exr(f) for f in file_names])
    with ThreadPoolExecutor(max_workers=int(threads)) as pool:
        return np.stack(list(pool.map(get_depthmap_from_exr, file_names)))


def _depth_array(depth):
    return get_depthmap_from_exr(depth) if isinstance(depth, str) else depth


def prune_pts(pt_array, depth_map_name, crop=False, crop_params=None):
    """run_proj.py:33-46 -- depth visibility (variant A) and optional crop test + remap.

    depth_map_name may be a file name (see get_depthmap_from_exr) or the (H,W) float32 array
    itself.  With crop=True the caller's pt_array is rewritten in crop coordinates, exactly as
    the reference does, and returned as the second value."""
    depth_map = _depth_array(depth_map_name)
    depth_truth, depth_pts = pt2render.depth_visibility_array(pt_array, depth_map)
    if crop:
        crop_truth, crop_pts = pt2render.crop_visibility_array(pt_array, crop_params)
    else:
        crop_truth = torch.ones_like(depth_truth, dtype=torch.float64) if dv.is_tensor(depth_truth) \
            else np.ones(depth_truth.shape)
        crop_pts = depth_pts
    return crop_truth * depth_truth, crop_pts


def project_prune(model_pts, cams, depth_maps, crop_params=None, variant="A", error_relaxation=None,
                  want_parts=False):
    """One-launch batched get_3d_to_2d + prune_pts over n_pose cameras.

    model_pts (N,3); cams (n_pose,3,4); depth_maps (H,W) shared or (n_pose,H,W);
    crop_params None, one [top,bottom,left,right] or (n_pose,4).
    Returns (truth (n_pose,N) bool, pts (n_pose,3,N) f64[, depth_truth, crop_truth])."""
    as_tensor = dv.is_tensor(model_pts)
    pts = dv.to_dev(model_pts, torch.float64)
    cams_t = dv.to_dev(cams, torch.float64).reshape(-1, 3, 4)
    depth = dv.to_dev(depth_maps, torch.float32)
    per_pose = 1 if depth.dim() == 3 else 0
    h, w = depth.shape[-2:]
    n, n_pose = pts.shape[0], cams_t.shape[0]
    if per_pose and depth.shape[0] != n_pose:
        raise ValueError("need one depth map per pose or a single shared one")
    crop_t, crop_per_pose = None, 0
    if crop_params is not None:
        crop_t = dv.rows(crop_params, 4, torch.int32)
        crop_per_pose = 1 if crop_t.shape[0] > 1 else 0
        if crop_per_pose and crop_t.shape[0] != n_pose:
            raise ValueError("need one crop per pose or a single shared one")
    if error_relaxation is None:
        error_relaxation = 0.15 if variant == "A" else 0.05
    out = dv.empty((n_pose, 3, n), torch.float64)
    truth = dv.empty((n_pose, n), torch.uint8)
    td = dv.empty((n_pose, n), torch.uint8) if want_parts else None
    tc = dv.empty((n_pose, n), torch.uint8) if want_parts else None
    _abi.check(_abi.lib.ispa_project_prune(
        dv.ptr(pts), n, dv.ptr(cams_t), n_pose, dv.ptr(depth), per_pose, h, w, min(VIS_YLIM, h),
        min(VIS_XLIM, w), float(error_relaxation), _abi.VIS_A if variant == "A" else _abi.VIS_B,
        dv.ptr(crop_t), crop_per_pose, dv.ptr(out), dv.ptr(truth), dv.ptr(td), dv.ptr(tc), dv.stream_ptr()))
    res = [dv.back(truth.bool(), as_tensor), dv.back(out, as_tensor)]
    if want_parts:
        res += [dv.back(td.bool(), as_tensor), dv.back(tc.bool(), as_tensor)]
    return tuple(res)


def spatial_order(points, bits=10, curve="hilbert"):
    """Permutation that puts a point set in the order of a space-filling curve through its bounding box
    (`curve` = "hilbert", or "morton" for the Z-curve).

    Optional one-time preprocessing of a model's surface samples (the reference draws them in random
    order, run_proj.py:15-22): points that are neighbours in space become neighbours in the array, so
    the lanes of a wave project onto neighbouring pixels, the depth gathers and LDS atomics of the kernels
    coalesce, and the tiled z-buffer pass (zbuffer.project_zbuffer) can skip, per screen tile, the chunks
    of consecutive points that project elsewhere.  The Hilbert curve has no jumps, so a run of consecutive
    points is a more compact blob than on the Z-curve.  Per-point outputs come back in the order given, so
    results for `points[perm]` are the results for `points` permuted the same way; only "last writer wins"
    splats (make_disparity) depend on the order itself, as they do in the reference."""
    p = np.asarray(points, dtype=np.float64)
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError("points must be (N,3)")
    if curve not in ("hilbert", "morton"):
        raise ValueError("curve must be 'hilbert' or 'morton'")
    if len(p) == 0:
        return np.zeros(0, dtype=np.int64)
    lo, hi = p.min(0), p.max(0)
    scale = (2 ** bits - 1) / np.where(hi > lo, hi - lo, 1.0)
    g = np.clip(((p - lo) * scale).astype(np.int64), 0, 2 ** bits - 1)
    if curve == "hilbert":
        g = _hilbert_transpose(g, bits)
    code = np.zeros(len(p), dtype=np.int64)
    for b in range(bits):                            # interleave the bits of the three axes
        for axis in range(3):
            # Hilbert: axis 0 carries the most significant bit of each triple
            shift = 3 * b + (2 - axis if curve == "hilbert" else axis)
            code |= ((g[:, axis] >> b) & 1) << shift
    return np.argsort(code, kind="stable")


def _hilbert_transpose(grid, bits):
    """Skilling's axes-to-transpose transform (AIP Conf. Proc. 707, 2004), vectorised over points: integer
    grid coordinates (N,3) -> the 'transposed' Hilbert index, whose bit-interleaving is the curve position."""
    x = grid.copy()
    q = 1 << (bits - 1)
    while q > 1:                                     # inverse undo of the excess work
        mask = q - 1
        for i in range(3):
            flip = (x[:, i] & q) != 0
            x[flip, 0] ^= mask                       # invert the low bits of axis 0
            t = (x[:, 0] ^ x[:, i]) & mask           # else exchange the low bits of axis 0 and axis i
            t[flip] = 0
            x[:, 0] ^= t
            x[:, i] ^= t
        q >>= 1
    for i in range(1, 3):                            # Gray encode
        x[:, i] ^= x[:, i - 1]
    t = np.zeros(len(x), dtype=np.int64)
    q = 1 << (bits - 1)
    while q > 1:
        sel = (x[:, 2] & q) != 0
        t[sel] ^= q - 1
        q >>= 1
    for i in range(3):
        x[:, i] ^= t
    return x
