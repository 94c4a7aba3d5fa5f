"""Oracle (test infrastructure): filename codec, skeleton densifier, keypoint -> grid index.

Restates (from source text; the modules need cv2 / scipy.io / OpenEXR to import)
  data_creator/utils.py:192-251,352-374
  data_creator/global_variables.py:55-71 (the tables are pinned against the imported module
      through tests/golden/tables.json)
  data_creator/make_dense_pointset_syn.py:27-67
  data_creator/make_dense_pointset.py:70-77 (2-D linspace only)
  classifier_code/train_test_scripts/utils/io_utils.py:20-31,126-148
  classifier_code/train_test_scripts/utils/proxy_renderer.py:50-58,100-151
  classifier_code/train_test_scripts/multi_view_train.py:164,177 and
  classifier_code/train_test_scripts/arch/eccv_net/eccv_net.py:282,210
  ucn_code/lib/layers/hard_negative_processing_layer.py:44-58,188-189
"""
import numpy as np

from . import geometry as g

# global_variables.py:58-71 : (bones, side tags, seat corner ids)
SKELETON_MAP = {
    "chair": ([[4, 0], [5, 1], [6, 2], [7, 3], [4, 5], [5, 6], [6, 7], [7, 4], [7, 8], [8, 9],
               [9, 6], [10, 12], [11, 13]],
              ["L", "R", "R", "L", "C", "R", "C", "L", "L", "C", "R", "C", "C"], [4, 5, 6, 7]),
    "swivelchair": ([[4, 0], [5, 1], [6, 2], [7, 3], [7, 8], [8, 9], [9, 10], [10, 7], [10, 11],
                     [11, 12], [12, 9], [13, 15], [14, 16]],
                    ["O", "O", "O", "O", "C", "R", "C", "L", "L", "C", "R", "C", "C"],
                    [7, 8, 9, 10]),
    "bed": ([[3, 1], [8, 6], [7, 5], [2, 0], [3, 8], [8, 7], [7, 2], [2, 3], [2, 4], [4, 9], [9, 7],
             [13, 14], [10, 15], [11, 16], [0, 1], [1, 6], [6, 5], [5, 0], [10, 12], [11, 13]],
            ["L", "R", "RR", "LL", "C", "R", "C", "L", "L", "C", "R", "LL", "C", "RR", "LL", "C",
             "RR", "O", "C", "C"], [3, 8, 7, 2]),
    "table": ([[0, 4], [3, 7], [2, 6], [1, 5], [0, 3], [3, 2], [2, 1], [1, 0], [8, 10], [9, 11]],
              ["L", "R", "R", "L", "C", "R", "C", "L", "C", "C"], [0, 3, 2, 1]),
    "sofa": ([[3, 1], [10, 8], [11, 7], [4, 0], [3, 10], [10, 9], [9, 2], [2, 3], [2, 6], [6, 13],
              [13, 9], [15, 14], [6, 4], [4, 5], [5, 3], [13, 11], [11, 12], [12, 10], [0, 1],
              [1, 8], [8, 7], [7, 0], [15, 16], [16, 17]],
             ["L", "R", "RR", "LL", "C", "R", "C", "L", "L", "C", "R", "C", "L", "L", "L", "R", "R",
              "R", "LL", "C", "RR", "O", "C", "C"], [3, 10, 9, 2]),
    "car": ([[3, 16], [4, 17], [21, 34], [22, 35], [1, 3], [3, 4], [4, 6], [6, 17], [17, 16], [16, 1],
             [19, 21], [21, 22], [22, 24], [24, 35], [35, 34], [34, 19], [1, 19], [3, 21], [4, 22],
             [6, 24]],
            ["LL", "LL", "RR", "RR", "L", "L", "L", "LL", "LL", "LL", "R", "R", "R", "RR", "RR", "RR",
             "C", "C", "C", "C"], [3, 4, 21, 22]),
}
SKELETON_MAP["diningtable"] = SKELETON_MAP["table"]

# utils.py:199-251 : per class, the column pairs averaged (nanmean) into extra anchors.
_MIDPOINTS = {
    "chair": [(4, 5), (5, 6), (6, 7), (7, 4)],
    "swivelchair": [(7, 8), (8, 9), (9, 10), (10, 7)],
    "table": [(0, 1), (1, 2), (2, 3), (0, 3)],
    "bed": [(3, 8), (7, 8), (7, 2), (2, 3), (0, 1), (1, 6), (5, 6)],
    "sofa": [(1, 8), (3, 10), (2, 9), (6, 13)],
}


def get_params_from_syn_file(file_name):
    """utils.py:359-370 == io_utils.py:20-31: pose and crop from a render file name.

    '<synset>_<md5>_a<az>_e<el>_t<tilt>_d<rho>_<top>_<bottom>_<left>_<right>_.jpg' with '@'
    for the decimal point; the tilt is NEGATED.
    """
    f = file_name.split("/")[-1].split("_")

    def num(s):
        return float(s[1:].replace("@", "."))
    return {"model_type": f[0], "model_md5": f[1], "azimuth": num(f[2]), "elevation": num(f[3]),
            "tilt": -num(f[4]), "rho": num(f[5]),
            "crop_params": [int(f[6]), int(f[7]), int(f[8]), int(f[9])]}


def img_to_depth_file(img):
    """utils.py:372-374."""
    return "_".join(img.split("_")[:-5]) + ".png1.exr"


def name_version(value):
    """io_utils.py:147-148."""
    return ("%3.3f" % value).replace(".", "@")


def render_file_name(d):
    """io_utils.py:132-136 without the directory part (tilt negated back)."""
    c = d["crop_params"]
    return (d["model_type"] + "_" + d["model_md5"] + "_a" + name_version(d["azimuth"]) + "_e"
            + name_version(d["elevation"]) + "_t" + name_version(-d["tilt"]) + "_d"
            + name_version(d["rho"]) + "_%d_%d_%d_%d_.jpg" % (c[0], c[1], c[2], c[3]))


def depth_map_name(d):
    """io_utils.py:126-130 without the directory part."""
    return (d["model_type"] + "_" + d["model_md5"] + "_a" + name_version(d["azimuth"]) + "_e"
            + name_version(d["elevation"]) + "_t" + name_version(-d["tilt"]) + "_d"
            + name_version(d["rho"]) + ".png1.exr")


def get_3d_keypoints(pt_1, pt_2, n):
    """utils.py:352-357."""
    return np.stack([np.linspace(pt_1[d], pt_2[d], n) for d in range(3)])


def get_2d_keypoints(pt_1, pt_2, n):
    """utils.py:192-196."""
    return np.stack([np.linspace(pt_1[d], pt_2[d], n) for d in range(2)])


def get_additional_coords_keypoint_5(coords, class_name, dim=3):
    """utils.py:199-251: append class-specific midpoints (nanmean of two anchors)."""
    pairs = _MIDPOINTS.get(class_name)
    if pairs is None:
        return coords
    extra = np.zeros((dim, len(pairs)))
    for n, (a, b) in enumerate(pairs):
        extra[:, n] = np.nanmean(np.stack([coords[:, a], coords[:, b]], axis=1), axis=1)
    return np.concatenate([coords, extra], axis=1)


def dense_skeleton_3d(anchors_k3, class_name, per_bone=10):
    """make_dense_pointset_syn.py:19,27-40: anchors (K,3) -> (3, per_bone * n_bones)."""
    coords = get_additional_coords_keypoint_5(np.asarray(anchors_k3, dtype=np.float64).T, class_name)
    bones, tags, _ = SKELETON_MAP[class_name]
    cols = []
    for (i, j), tag in zip(bones, tags):
        cols.append(np.zeros((3, per_bone)) if tag == "O"
                    else get_3d_keypoints(coords[:, i], coords[:, j], per_bone))
    return np.concatenate(cols, axis=1)


def dense_skeleton_2d(coords_2d, class_name, per_bone=10):
    """make_dense_pointset.py:70-77 (linspace part): (>=2, K') image coords -> (2, 10*n_bones)."""
    bones, tags, _ = SKELETON_MAP[class_name]
    cols = []
    for (i, j), tag in zip(bones, tags):
        cols.append(np.zeros((2, per_bone)) if tag == "O"
                    else get_2d_keypoints(coords_2d[:, i], coords_2d[:, j], per_bone))
    return np.concatenate(cols, axis=1)


def make_dense_keypoints_core(anchors_k3, class_name, params, depth_map, crop_params):
    """make_dense_pointset_syn.py:54-67 for one render: (2, 10*n_bones), invisible columns zeroed."""
    pts3d = dense_skeleton_3d(anchors_k3, class_name).T.copy()
    key_2d = g.get_3d_to_2d(pts3d, params)
    truth, key_2d = g.prune_pts(key_2d, depth_map, crop=True, crop_params=crop_params)
    key_2d[:, ~truth.astype(bool)] = 0
    return key_2d[:2]


def get_pt(anchors_k3, file_name, depth_map):
    """proxy_renderer.py:50-58: (1,K,3) = [u224, v224, visible] through variant-B pruning."""
    anchors = np.asarray(anchors_k3, dtype=np.float64)
    truth, crop_pts = g.get_render_array(get_params_from_syn_file(file_name), anchors, depth_map)
    out = np.zeros((1, anchors.shape[0], 3))
    out[0, :, :2] = crop_pts[:2].T
    out[0, :, 2] = truth
    return out


def knn_inds_to_grid(knn_xy_224, grid=28, image=224):
    """multi_view_train.py:164,177 + eccv_net.py:282: clip to [0,223], /(224/28), truncate to int64."""
    clipped = np.clip(np.asarray(knn_xy_224, dtype=np.float64), 0, image - 1)
    return (clipped / (float(image) / grid)).astype(np.int64)


def ind2coord(ind_one_based, width, downsampling=4):
    """hard_negative_processing_layer.py:44-58,188-189: 1-based flat KNN index -> image (x, y)."""
    ind = np.asarray(ind_one_based) - 1
    x = ind % width
    y = np.floor(ind / width)
    return np.stack([x, y], axis=-1) * downsampling


def normalize_pose_delta(delta):
    """proxy_renderer.py:138-151 (in place on a copy)."""
    d = np.array(delta, dtype=np.float64, copy=True)
    for col, half in ((0, 180), (1, 90), (2, 180)):
        c = d[:, col].copy()
        d[c >= half, col] -= 2 * half
        d[c < -half, col] += 2 * half
    return d


def nearest_view(view_params, param_np):
    """proxy_renderer.py:101-106: index of the stored render nearest to the requested pose."""
    d = normalize_pose_delta(np.asarray(param_np, dtype=np.float64) - np.asarray(view_params))
    d[:, 1:] /= 8.0
    return int(np.argmin(np.linalg.norm(d, axis=1)))


def pck_counts(knn_inds_one_based, correspondences, width, radii, downsampling=4):
    """hard_negative_processing_layer.py:188-190,218-234: per batch item and radius, the number of
    valid ground-truth correspondences (flag column 4 == 1) whose 1-NN lands within `radius`
    image pixels of the ground-truth position (columns 2:4)."""
    knn = np.asarray(knn_inds_one_based)
    corr = np.asarray(correspondences)
    xy = ind2coord(knn, width, downsampling)                 # (B, k, n, 2) image coordinates
    out = np.zeros((len(radii), corr.shape[0]))
    for n in range(corr.shape[0]):
        ones = corr[n, :, 4] == 1
        d = np.sqrt(np.sum((xy[n, 0, ones] - corr[n, ones, 2:4]) ** 2, axis=1))
        for i, rad in enumerate(radii):
            out[i, n] = np.sum(d < rad)
    return out


def hard_negatives(knn_inds_one_based, correspondences, width, thresh=4, downsampling=4):
    """hard_negative_processing_layer.py:188-192,246-250: per valid ground-truth correspondence (flag column 4 ==
    1), whether its 1-NN lies farther than `thresh` from the ground truth in FEATURE coordinates
    (cfg.NEGATIVE_MINING_PIXEL_THRESH, ucn_code/lib/config.py:80), and the image coordinate of every row's 1-NN
    (line 189).  Returns (mask (B,n) bool, False on rows that are not valid; knn_xy (B,n,2))."""
    knn = np.asarray(knn_inds_one_based)
    corr = np.asarray(correspondences)
    feat_knn = ind2coord(knn, width, 1)                       # (B, k, n, 2) feature coordinates
    img_knn = feat_knn * downsampling                         # _transform_coord_feat2im
    feat_gt = corr[:, :, 2:4] / downsampling                  # _transform_coord_im2feat
    mask = np.zeros(corr.shape[:2], dtype=bool)
    for n in range(corr.shape[0]):
        ones = corr[n, :, 4] == 1
        feat_dist = np.sqrt(np.sum((feat_knn[n, 0, ones] - feat_gt[n, ones]) ** 2, axis=1))
        mask[n, ones] = feat_dist > thresh
    return mask, img_knn[:, 0]
