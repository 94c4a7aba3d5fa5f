"""Dense skeleton keypoints: anchors -> bones -> 10 points per bone -> projected per render.

Drop-in for data_creator/make_dense_pointset_syn.py:16-67 and for the data-in/data-out core of
data_creator/make_dense_pointset.py:38-99 (real images: bones, then the three pruning passes of
data_creator/utils.py:274-348 as one batched device launch), with the helpers of
data_creator/utils.py:192-270, 352-357 and the tables of data_creator/global_variables.py:55-75.
"""
import numpy as np
import torch

from . import _abi, run_proj
from . import _device as dv

# (bones as anchor-index pairs, side tag per bone, seat corners) -- global_variables.py:58-71
skeleton_map = {
    "chair": ([[4, 0], [5, 1], [6, 2], [7, 3], [4, 5], [5, 6], [6, 7], [7, 4], [7, 8], [8, 9], [9, 6],
               [10, 12], [11, 13]],
              list("LRRLCRCLLCRCC"), [4, 5, 6, 7]),
    "swivelchair": ([[4, 0], [5, 1], [6, 2], [7, 3], [7, 8], [8, 9], [9, 10], [10, 7], [10, 11], [11, 12],
                     [12, 9], [13, 15], [14, 16]],
                    list("OOOOCRCLLCRCC"), [7, 8, 9, 10]),
    "bed": ([[3, 1], [8, 6], [7, 5], [2, 0], [3, 8], [8, 7], [7, 2], [2, 3], [2, 4], [4, 9], [9, 7], [13, 14],
             [10, 15], [11, 16], [0, 1], [1, 6], [6, 5], [5, 0], [10, 12], [11, 13]],
            "L R RR LL C R C L L C R LL C RR LL C RR O C C".split(), [3, 8, 7, 2]),
    "table": ([[0, 4], [3, 7], [2, 6], [1, 5], [0, 3], [3, 2], [2, 1], [1, 0], [8, 10], [9, 11]],
              list("LRRLCRCLCC"), [0, 3, 2, 1]),
    "sofa": ([[3, 1], [10, 8], [11, 7], [4, 0], [3, 10], [10, 9], [9, 2], [2, 3], [2, 6], [6, 13], [13, 9],
              [15, 14], [6, 4], [4, 5], [5, 3], [13, 11], [11, 12], [12, 10], [0, 1], [1, 8], [8, 7], [7, 0],
              [15, 16], [16, 17]],
             "L R RR LL C R C L L C R C L L L R R R LL C RR O C C".split(), [3, 10, 9, 2]),
    "car": ([[3, 16], [4, 17], [21, 34], [22, 35], [1, 3], [3, 4], [4, 6], [6, 17], [17, 16], [16, 1], [19, 21],
             [21, 22], [22, 24], [24, 35], [35, 34], [34, 19], [1, 19], [3, 21], [4, 22], [6, 24]],
            "LL LL RR RR L L L LL LL LL R R R RR RR RR C C C C".split(), [3, 4, 21, 22]),
}
skeleton_map["diningtable"] = (skeleton_map["table"][0], list(skeleton_map["table"][1]), skeleton_map["table"][2])
skeleton_size = {"chair": 10, "swivelchair": 13, "table": 8, "sofa": 14, "bed": 10}   # global_variables.py:55

# extra anchors = midpoint (nanmean) of two annotated anchors -- utils.py:199-251
_midpoint_pairs = {
    "chair": ((4, 5), (5, 6), (6, 7), (7, 4)),
    "swivelchair": ((7, 8), (8, 9), (9, 10), (10, 7)),
    "table": ((0, 1), (1, 2), (2, 3), (0, 3)),
    "bed": ((3, 8), (7, 8), (7, 2), (2, 3), (0, 1), (1, 6), (5, 6)),
    "sofa": ((1, 8), (3, 10), (2, 9), (6, 13)),
}
POINTS_PER_BONE = 10

# dense-point columns tested against the seat outline -- global_variables.py:73-74
leg_range = {"chair": range(40), "swivelchair": range(40), "bed": list(range(40)) + list(range(110, 140)),
             "table": range(40), "diningtable": range(40), "sofa": list(range(40)) + list(range(110, 120)),
             "car": list(range(40)) + list(range(70, 100)) + list(range(130, 160))}
img_size = (224, 224)                                    # global_variables.py:75
_pascal_extra = {"chair": 4, "swivelchair": 4, "diningtable": 4, "bed": 7, "sofa": 4}   # utils.py:253-270
LINE_THICKNESS = 5                                       # utils.py:314


def _linspace_rows(pt_1, pt_2, n, dim):
    out = np.zeros((dim, n))
    for d in range(dim):
        out[d, :] = np.linspace(pt_1[d], pt_2[d], n)
    return out


def get_2d_keypoints(pt_1, pt_2, n):
    """utils.py:192-196."""
    return _linspace_rows(pt_1, pt_2, n, 2)


def get_3d_keypoints(pt_1, pt_2, n):
    """utils.py:352-357."""
    return _linspace_rows(pt_1, pt_2, n, 3)


def get_additional_coords_keypoint_5(img_coords, class_name, dim=3):
    """utils.py:199-251 -- (dim,K) anchors -> (dim,K+extra) with the class's midpoints appended."""
    pairs = _midpoint_pairs.get(class_name)
    if not pairs:
        return img_coords
    extra = np.zeros((dim, len(pairs)))
    for col, (a, b) in enumerate(pairs):
        extra[:, col] = np.nanmean(np.concatenate([img_coords[:, a:a + 1], img_coords[:, b:b + 1]], 1), 1)
    return np.concatenate([img_coords, extra], 1)


def dense_skeleton(coords, class_name, dim=3):
    """make_dense_pointset_syn.py:33-40 / make_dense_pointset.py:70-77: (dim,K') -> (dim, 10*n_bones);
    bones tagged 'O' contribute zeros."""
    bones, tags, _ = skeleton_map[class_name]
    cols = [np.zeros((dim, POINTS_PER_BONE)) if tag == "O"
            else _linspace_rows(coords[:, i], coords[:, j], POINTS_PER_BONE, dim)
            for (i, j), tag in zip(bones, tags)]
    return np.concatenate(cols, 1) if cols else np.zeros((dim, 0))


def dense_keypoints_for_renders(anchors, class_name, params, depth_maps, crop_params):
    """make_dense_pointset_syn.py:54-67 batched over renders.

    anchors (K,3) as in anchors.txt; params (n,4); depth_maps (n,540,960) f32; crop_params (n,4).
    Returns (n, 2, 10*n_bones) float64 crop coordinates with invisible points zeroed."""
    coords = get_additional_coords_keypoint_5(np.asarray(anchors, dtype=np.float64).T, class_name)
    pts3d = np.ascontiguousarray(dense_skeleton(coords, class_name).T)
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    cams = run_proj.cameras_from_params(params)
    depth = np.asarray(depth_maps)
    if depth.ndim == 2:
        depth = depth[None]
    truth, key_2d = run_proj.project_prune(pts3d, cams, depth, crop_params)
    key_2d = key_2d[:, :2, :].copy()
    key_2d[~np.broadcast_to(truth[:, None, :], key_2d.shape)] = 0
    return key_2d


def get_additional_coords_pascal(img_coords, class_name, dim=3):
    """utils.py:253-270 -- zero columns in the places of keypoint-5's midpoints."""
    extra = _pascal_extra.get(class_name)
    if extra is None:
        return img_coords
    return np.concatenate([img_coords, np.zeros((dim, extra))], 1)


# ------------------------------------------------------------------ pruning on real images
def _facing(img_coords, seat_info):
    """utils.py:293-296: x component of (front seat edge centre - back seat edge centre)."""
    front = (img_coords[:2, seat_info[0]] + img_coords[:2, seat_info[1]]) / 2
    back = (img_coords[:2, seat_info[2]] + img_coords[:2, seat_info[3]]) / 2
    return (front - back)[0]


def _launch_prune(all_coords, flags, polygon, segments, n_seg):
    """all_coords (B,2,N) float64 numpy (edited in place) or device tensor (edited in place)."""
    as_tensor = dv.is_tensor(all_coords)
    b, _, n = all_coords.shape
    if as_tensor and not (all_coords.is_cuda and all_coords.dtype == torch.float64 and all_coords.is_contiguous()):
        raise ValueError("a tensor of dense points must be a contiguous float64 device tensor (it is edited in place)")
    c = all_coords if as_tensor else dv.to_dev(all_coords, torch.float64)
    f = dv.to_dev(flags, torch.uint8)
    poly = dv.to_dev(polygon, torch.int32)
    seg = dv.to_dev(segments, torch.int32)
    ns = dv.to_dev(n_seg, torch.int32)
    pruned = dv.empty((b, n), torch.uint8)
    _abi.check(_abi.lib.ispa_prune_points_2d(dv.ptr(c), b, n, dv.ptr(f), dv.ptr(poly), poly.shape[1],
                                             dv.ptr(seg) if seg.shape[1] else None,
                                             dv.ptr(ns) if seg.shape[1] else None, seg.shape[1], LINE_THICKNESS,
                                             img_size[0], img_size[1], dv.ptr(pruned), dv.stream_ptr()))
    if not as_tensor:
        all_coords[...] = c.cpu().numpy()
    return all_coords, dv.back(pruned, as_tensor)


def _prune_plan(img_coords, n_points, skeleton, skele_info, seat_info, legs, surface=True, direction=True,
                surface_dir=False):
    """Host side of the three passes for one image: which points take which test, the seat outline
    (truncated to int32 as np.array(..., dtype=np.int32) does) and the bones to draw."""
    flags = np.zeros(n_points, np.uint8)
    polygon = np.array([img_coords[:2, seat_info[c]] for c in range(4)], dtype=np.int32)
    segments = []
    if surface:
        flags[np.asarray(list(legs), dtype=np.int64)] |= _abi.PRUNE_POLYGON
    d = _facing(img_coords, seat_info) if (direction or surface_dir) else 0
    n = len(skeleton)
    if direction and d != 0:
        far, near = ("L", "R") if d > 0 else ("R", "L")
        for i in range(n):
            if far in skele_info[i]:
                a, c = skeleton[i]
                segments.append(list(img_coords[:2, a].astype(int)) + list(img_coords[:2, c].astype(int)))
        for i in range(n):
            if near in skele_info[i] or "C" in skele_info[i]:
                # skeleton.index(bone): the FIRST bone with these end points takes the test
                first = skeleton.index(skeleton[i])
                flags[first * 10 + 1:first * 10 + 9] |= _abi.PRUNE_BAND
    if surface_dir and d != 0:
        tag = "RR" if d > 0 else "LL"
        for i in range(n):
            if tag in skele_info[i]:
                first = skeleton.index(skeleton[i])
                flags[first * 10:first * 10 + 10] |= _abi.PRUNE_DROP
    return flags, polygon, segments


def _run_plans(all_coords, plans):
    b = len(plans)
    max_seg = max((len(p[2]) for p in plans), default=0)
    seg = np.zeros((b, max_seg, 4), np.int32)
    n_seg = np.zeros(b, np.int32)
    for i, p in enumerate(plans):
        n_seg[i] = len(p[2])
        if p[2]:
            seg[i, :len(p[2])] = np.asarray(p[2], dtype=np.int32)
    return _launch_prune(all_coords, np.stack([p[0] for p in plans]), np.stack([p[1] for p in plans]), seg, n_seg)


def surface_pruned(all_coords, img_coords, seat_info, k=range(40)):
    """utils.py:274-290: zero (in place) the points of columns `k` inside or on the seat outline."""
    plan = _prune_plan(img_coords, all_coords.shape[1], [], [], seat_info, k, direction=False)
    _run_plans(all_coords[None], [plan])
    return all_coords


def dir_pruned(all_coords, img_coords, skeleton, seat_info, skele_info):
    """utils.py:292-325: zero (in place) points 1..8 of the near-side and centre bones that lie under
    the 5-pixel lines of the far-side bones."""
    plan = _prune_plan(img_coords, all_coords.shape[1], skeleton, skele_info, seat_info, (), surface=False)
    _run_plans(all_coords[None], [plan])
    return all_coords


def surface_dir_prune(all_coords, img_coords, skeleton, skele_info, seat_info):
    """utils.py:327-348: zero (in place) the bones tagged RR (object faces right) or LL (faces left)."""
    plan = _prune_plan(img_coords, all_coords.shape[1], skeleton, skele_info, seat_info, (), surface=False,
                       direction=False, surface_dir=True)
    _run_plans(all_coords[None], [plan])
    return all_coords


def make_dense_keypoints(sparse_keypoints, class_name, dataset="keypoint-5"):
    """make_dense_pointset.py:43-99 for a batch of images, data in / data out.

    sparse_keypoints (B,K,3) rows [x, y, present] as the sparse_keypoints/*.npy files hold them.
    Returns (all_coords (B,2,10*n_bones) float64, saved (B,) bool): saved[b] is False where the
    reference writes no file (no bone has both end points annotated); those rows are zero.
    The three pruning passes run as ONE device launch over the batch."""
    skeleton, skele_info, seat_info = skeleton_map[class_name]
    sparse = np.asarray(sparse_keypoints)
    if sparse.ndim != 3 or sparse.shape[2] != 3:
        raise ValueError("sparse_keypoints must be (B,K,3)")
    n_points = POINTS_PER_BONE * len(skeleton)
    out = np.zeros((sparse.shape[0], 2, n_points))
    saved = np.zeros(sparse.shape[0], bool)
    plans = []
    for b in range(sparse.shape[0]):
        img_coords = np.clip(sparse[b], 0, 223).astype(int).T
        if dataset == "keypoint-5":
            img_coords = get_additional_coords_keypoint_5(img_coords, class_name)
            act = list(skele_info)
        else:
            img_coords = get_additional_coords_pascal(img_coords, class_name)
            present = img_coords[2, :]
            act = [tag if present[i] == 1 and present[j] == 1 else "O" for (i, j), tag in zip(skeleton, skele_info)]
        saved[b] = any(tag != "O" for tag in act)
        if not saved[b]:
            plans.append((np.zeros(n_points, np.uint8), np.zeros((4, 2), np.int32), []))
            continue
        cols = [np.zeros((2, POINTS_PER_BONE)) if tag == "O"
                else get_2d_keypoints(img_coords[:, i], img_coords[:, j], POINTS_PER_BONE)
                for (i, j), tag in zip(skeleton, act)]
        out[b] = np.concatenate(cols, 1)
        plans.append(_prune_plan(img_coords, n_points, skeleton, skele_info, seat_info, leg_range[class_name],
                                 surface_dir=class_name in ("bed", "sofa")))
    if len(plans):
        _run_plans(out, plans)
    return out, saved
