"""Oracle (test infrastructure): camera model, projection, crop remap, depth visibility.

Restates, in float64 numpy,
  data_creator/render_wt_pt_proj/pt2render.py:6-87      (variant A)
  classifier_code/train_test_scripts/utils/pt2render.py:6-85  (variant B)
  data_creator/render_wt_pt_proj/variables.py:3-9
  data_creator/render_wt_pt_proj/run_proj.py:24-46
  classifier_code/train_test_scripts/utils/io_utils.py:97-124
Pinned by tests/golden/geometry_*.npz (generated from the imported reference).
"""

import numpy as np

# variables.py:3-9 -- integer arrays in the reference; products promote to float64.
WORLD_MATRIX = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
INTRINSIC_CAMERA_MATRIX = np.array([[1050, 0, 480], [0, 1050, 270], [0, 0, 1]])

IMG_H = 540      # pt2render.py:47  (y < 540)
IMG_XLIM = 950   # pt2render.py:47  (x < 950, although the image is 960 wide)
CROP_SIDE = 224  # pt2render.py:28,33


def rot_three_angles(yaw, pitch, roll):
    """pt2render.py:58-73: R = (Rz(yaw) Ry(-pitch) Rx(roll))^T with rows rotated up by one."""
    a = np.deg2rad(yaw)
    p = np.deg2rad(-pitch)
    r = np.deg2rad(roll)
    rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    m = np.matmul(rz, np.matmul(ry, rx)).T
    return np.roll(m, -1, axis=0)


def ex_camera_matrix(yaw, pitch, roll, rho):
    """pt2render.py:76-84: [R | -(0,0,rho)] with rows 1 and 2 negated."""
    e = np.zeros((3, 4))
    e[:, :3] = rot_three_angles(yaw, pitch, roll)
    e[:, 3] = -np.array([0, 0, rho])
    e[1:, :] = -e[1:, :]
    return e


def get_camera_matrix(intrinsic, extrinsic, world):
    """pt2render.py:86-87: P = K . E . W."""
    return np.matmul(np.matmul(intrinsic, extrinsic), world)


def camera_from_params(params):
    """run_proj.py:25-27: params = [azimuth, elevation, tilt, rho] -> P (3,4)."""
    e = ex_camera_matrix(params[0], params[1], params[2], params[3])
    return get_camera_matrix(INTRINSIC_CAMERA_MATRIX, e, WORLD_MATRIX)


def get_render_location(pt_array, cam):
    """pt2render.py:6-13: (N,3) points -> (3,N) rows [round(u), round(v), z_cam].

    numpy's matmul hands this 3x4 by 4xN product to dgemm, whose accumulation order is
    the BLAS build's business; oracle.cnative.project_fma fixes the order (k-sequential
    fma chain from 0, which is what OpenBLAS did in the build container) and is the
    bit-exact twin of the HIP kernel.  The two agree exactly on u, v and to 1e-15 on z.
    """
    homog = np.concatenate([np.asarray(pt_array, dtype=np.float64).T,
                            np.ones((1, len(pt_array)))], axis=0)
    proj = np.matmul(cam, homog)
    z = proj[2].copy()
    out = np.round(proj / proj[2])
    out[2] = z
    return out


def crop_visibility_array(pt_render, crop_params, resize=True):
    """pt2render.py:18-38: strict crop test, then IN-PLACE remap of rows 0,1 to the 224 crop.

    crop_params = [top, bottom, left, right].  Returns (truth, pt_render) where
    pt_render is the caller's array, mutated.
    """
    top, bottom, left, right = crop_params
    inside = ((pt_render[1] > top) & (pt_render[1] < bottom)
              & (pt_render[0] > left) & (pt_render[0] < right))
    pt_render[0] -= left
    pt_render[1] -= top
    if resize:
        y_len = bottom - top
        x_len = right - left
        if y_len >= x_len:
            ratio = CROP_SIDE / float(y_len)
            pt_render[:2] *= ratio
            pt_render[0] += (CROP_SIDE - x_len * ratio) / 2
        else:
            ratio = CROP_SIDE / float(x_len)
            pt_render[:2] *= ratio
            pt_render[1] += (CROP_SIDE - y_len * ratio) / 2
        pt_render[:2] = np.round(pt_render[:2])
    return inside, pt_render


def _pixel_gate(pt_render):
    """pt2render.py:46-47: int() truncation of the rounded coords, 0<y<540, 0<x<950."""
    with np.errstate(invalid="ignore"):
        finite = np.isfinite(pt_render[0]) & np.isfinite(pt_render[1])
        y = np.where(finite, np.trunc(pt_render[1]), -1).astype(np.int64)
        x = np.where(finite, np.trunc(pt_render[0]), -1).astype(np.int64)
    gate = (y > 0) & (y < IMG_H) & (x > 0) & (x < IMG_XLIM)
    return y, x, gate


def depth_visibility_array_a(pt_render, depth_map, error_relaxation=0.15):
    """data_creator pt2render.py:43-55: |z - depth[y,x]| <= relax; array NOT pruned.

    depth_map is float32; z - depth promotes to float64 exactly as numpy scalars do.
    """
    y, x, gate = _pixel_gate(pt_render)
    yy = np.where(gate, y, 0)
    xx = np.where(gate, x, 0)
    d = depth_map[yy, xx].astype(np.float64)
    truth = gate & (np.abs(pt_render[2] - d) <= error_relaxation)
    return truth, pt_render


def depth_visibility_array_b(pt_render, depth_map, error_relaxation=0.05):
    """classifier_code pt2render.py:43-53: z <= min(depth[y-1:y+2, x-1:x+2] + relax).

    The sum depth + relax is evaluated in the depth map's dtype (float32 array plus a
    Python float stays float32), the minimum is taken in float32 and only the final
    comparison against the float64 z promotes.  Slices clip at the image border
    (only the bottom row can clip because y<540, x<950<960).  Returns the PRUNED array.
    """
    y, x, gate = _pixel_gate(pt_render)
    h, w = depth_map.shape
    relax = depth_map.dtype.type(error_relaxation)
    truth = np.zeros(pt_render.shape[1], dtype=bool)
    idx = np.nonzero(gate)[0]
    if idx.size:
        best = np.full(idx.size, np.inf, dtype=depth_map.dtype)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ry = y[idx] + dy
                rx = x[idx] + dx
                ok = (ry >= 0) & (ry < h) & (rx >= 0) & (rx < w)
                vals = depth_map[np.clip(ry, 0, h - 1), np.clip(rx, 0, w - 1)] + relax
                best = np.where(ok, np.minimum(best, vals), best)
        truth[idx] = pt_render[2, idx] <= best.astype(np.float64)
    return truth, pt_render[:, truth]


def depth_visibility_literal(pt_render, depth_map, variant, error_relaxation=None):
    """Per-point Python loop in the shape of the reference (small N only)."""
    n = pt_render.shape[1]
    truth = np.zeros(n, dtype=bool)
    if error_relaxation is None:
        error_relaxation = 0.15 if variant == "A" else 0.05
    for i in range(n):
        y, x = int(pt_render[1, i]), int(pt_render[0, i])
        if 0 < y < IMG_H and 0 < x < IMG_XLIM:
            if variant == "A":
                truth[i] = abs(pt_render[2, i] - depth_map[y, x]) <= error_relaxation
            else:
                truth[i] = pt_render[2, i] <= np.min(depth_map[y - 1:y + 2, x - 1:x + 2]
                                                     + error_relaxation)
    return truth


def get_3d_to_2d(model_pts, params):
    """run_proj.py:24-31."""
    return get_render_location(model_pts, camera_from_params(params))


def prune_pts(pt_array, depth_map, crop=False, crop_params=None):
    """run_proj.py:33-46 with the EXR read replaced by an ndarray argument (variant A)."""
    depth_truth, depth_pts = depth_visibility_array_a(pt_array, depth_map)
    if crop:
        crop_truth, crop_pts = crop_visibility_array(pt_array, crop_params)
    else:
        crop_truth = np.ones(depth_truth.shape)
        crop_pts = depth_pts
    return crop_truth * depth_truth, crop_pts


def get_render_array(file_dict, model_pts, depth_map):
    """io_utils.py:97-117 with the EXR read replaced by an ndarray argument (variant B)."""
    pts = get_3d_to_2d(model_pts, [file_dict["azimuth"], file_dict["elevation"],
                                   file_dict["tilt"], file_dict["rho"]])
    depth_truth, _ = depth_visibility_array_b(pts, depth_map)
    crop_truth, crop_pts = crop_visibility_array(pts, file_dict["crop_params"])
    return crop_truth * depth_truth, crop_pts


def py_slice_bounds(center, radius, size):
    """Bounds of the Python slice [center-radius : center+radius] on an axis of `size`."""
    def norm(v):
        if v < 0:
            v += size
            if v < 0:
                v = 0
        elif v > size:
            v = size
        return v
    lo, hi = norm(center - radius), norm(center + radius)
    return lo, max(lo, hi)


def get_pt_depth_map(render_array, size=(540, 960), radius=2):
    """io_utils.py:119-124: z splat, [v-2,v+2) x [u-2,u+2), point order, last writer wins."""
    out = np.zeros(size)
    for i in range(render_array.shape[1]):
        v, u = int(render_array[1, i]), int(render_array[0, i])
        out[v - radius:v + radius, u - radius:u + radius] = render_array[2, i]
    return out
