"""Dense skeleton keypoints: anchors -> bones -> 10 points per bone -> projected per render.

Drop-in for data_creator/make_dense_pointset_syn.py:16-67 and the linspace part of
data_creator/make_dense_pointset.py:70-77, with the helpers of data_creator/utils.py:192-251,
352-357 and the tables of data_creator/global_variables.py:55-71.
"""
import numpy as np

from . import run_proj

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
}
skeleton_map["diningtable"] = skeleton_map["table"]

# extra anchors = midpoint (nanmean) of two annotated anchors -- utils.py:199-251
_midpoint_pairs = {
    "chair": ((4, 5), (5, 6), (6, 7), (7, 4)),
    "swivelchair": ((7, 8), (8, 9), (9, 10), (10, 7)),
    "table": ((0, 1), (1, 2), (2, 3), (0, 3)),
    "bed": ((3, 8), (7, 8), (7, 2), (2, 3), (0, 1), (1, 6), (5, 6)),
    "sofa": ((1, 8), (3, 10), (2, 9), (6, 13)),
}
POINTS_PER_BONE = 10


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
    cams = np.stack([run_proj.camera_from_params(p) for p in params])
    depth = np.asarray(depth_maps)
    if depth.ndim == 2:
        depth = depth[None]
    truth, key_2d = run_proj.project_prune(pts3d, cams, depth, np.asarray(crop_params).reshape(len(params), 4))
    key_2d = key_2d[:, :2, :].copy()
    key_2d[~np.broadcast_to(truth[:, None, :], key_2d.shape)] = 0
    return key_2d
