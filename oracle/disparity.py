"""Oracle (test infrastructure): proxy optical flow ("disparity") of make_disparity.

Restates data_creator/make_disparity_syn.py:44-68 with the file reads replaced by
array arguments.  The module it lives in needs cv2 / trimesh / OpenEXR to import,
so this is a restatement from source text; tests/golden/make_golden.py re-types
the same loop on top of the imported reference primitives and pins it.
"""
import numpy as np

from . import geometry as g

FLOW_SIDE = 224   # make_disparity_syn.py:61
SPREAD = 2        # make_disparity_syn.py:63


def _hw(side):
    return (int(side), int(side)) if np.isscalar(side) else (int(side[0]), int(side[1]))


def splat_flow_literal(uv1, delta, side=FLOW_SIDE, k=SPREAD):
    """make_disparity_syn.py:61-68 (and get_of.py:54-60 with side = (540, 960)), literally: sequential overwrite
    with Python slices.

    uv1 (>=2, M) crop coordinates of the kept points, delta (>=2, M) their flow.
    A negative slice start wraps (Python semantics), which for 0 <= coordinate < k makes
    the slice empty: points within k pixels of the top/left border write nothing.
    """
    flow = np.zeros(_hw(side) + (2,))
    for i in range(uv1.shape[1]):
        pt = uv1[:2, i].astype(int)
        flow[pt[1] - k:pt[1] + k, pt[0] - k:pt[0] + k, 0] = delta[0, i]
        flow[pt[1] - k:pt[1] + k, pt[0] - k:pt[0] + k, 1] = delta[1, i]
    return flow


def splat_flow(uv1, delta, side=FLOW_SIDE, k=SPREAD):
    """Same result as splat_flow_literal, vectorised: per pixel the LAST point wins."""
    h, w = _hw(side)
    owner = np.full((h, w), -1, dtype=np.int64)
    m = uv1.shape[1]
    u = uv1[0].astype(int)
    v = uv1[1].astype(int)
    order = np.arange(m)
    ylo = np.empty(m, dtype=np.int64); yhi = np.empty(m, dtype=np.int64)
    xlo = np.empty(m, dtype=np.int64); xhi = np.empty(m, dtype=np.int64)
    for i in range(m):
        ylo[i], yhi[i] = g.py_slice_bounds(int(v[i]), k, h)
        xlo[i], xhi[i] = g.py_slice_bounds(int(u[i]), k, w)
    for dy in range(2 * k):
        for dx in range(2 * k):
            yy = ylo + dy
            xx = xlo + dx
            ok = (yy < yhi) & (xx < xhi)
            np.maximum.at(owner, (yy[ok], xx[ok]), order[ok])
    flow = np.zeros((h, w, 2))
    has = owner >= 0
    flow[has, 0] = delta[0, owner[has]]
    flow[has, 1] = delta[1, owner[has]]
    return flow


def make_disparity_core(key_3d, params, depth_map, crop_params, literal=False):
    """make_disparity_syn.py:44-68 for one rendered image.

    key_3d (N,3); params [az, el, tilt, rho]; depth_map (540,960) f32 of the FIRST pose;
    crop_params [top,bottom,left,right].  The second pose is (az+10, el+10, tilt, rho) and
    is pruned against the same depth map, but only the first pose's truth masks the points
    (make_disparity_syn.py:55).  Returns flow (224,224,2) float64.
    """
    p1 = list(params)
    pts1 = g.get_3d_to_2d(key_3d, p1)
    truth1, _ = g.prune_pts(pts1, depth_map, crop=True, crop_params=crop_params)
    p2 = [p1[0] + 10, p1[1] + 10, p1[2], p1[3]]
    pts2 = g.get_3d_to_2d(key_3d, p2)
    g.prune_pts(pts2, depth_map, crop=True, crop_params=crop_params)
    keep = truth1.astype(bool)
    uv2 = pts2[:, keep].copy()
    uv1 = pts1[:, keep].copy()
    delta = uv2 - uv1
    fn = splat_flow_literal if literal else splat_flow
    return fn(uv1, delta)


def demo_flow_core(key_3d, params, depth_map, step=(10, 0), literal=False):
    """data_creator/render_wt_pt_proj/get_of.py:38-62 for one view, the depth file replaced by an array.

    Both poses are pruned against the same depth image WITHOUT a crop (so the coordinates stay in render pixels),
    the points visible at both poses are kept (line 48), and their flow is splat into a map of the depth image's
    size (540 x 960 in the script)."""
    p1 = list(params)
    pts1 = g.get_3d_to_2d(key_3d, p1)
    truth1, _ = g.prune_pts(pts1, depth_map)
    p2 = [p1[0] + step[0], p1[1] + step[1], p1[2], p1[3]]
    pts2 = g.get_3d_to_2d(key_3d, p2)
    truth2, _ = g.prune_pts(pts2, depth_map)
    keep = truth1.astype(bool) & truth2.astype(bool)
    uv1 = pts1[:, keep].copy()
    delta = pts2[:, keep] - uv1
    fn = splat_flow_literal if literal else splat_flow
    return fn(uv1, delta, side=depth_map.shape)
