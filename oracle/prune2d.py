"""Oracle (test infrastructure): pruning of dense 2-D skeleton keypoints on real images.

Restates, from source text (the module needs cv2, which is not installed),
  data_creator/utils.py:253-270   get_additional_coords_pascal
  data_creator/utils.py:274-290   surface_pruned
  data_creator/utils.py:292-325   dir_pruned
  data_creator/utils.py:327-348   surface_dir_prune
  data_creator/make_dense_pointset.py:38-99   the per-image sequence around them
  data_creator/global_variables.py:73-75      leg_range, img_size

PARITY UNPINNED.  The reference delegates two decisions to OpenCV rasterisers that cannot be run
here and for which it holds no fixtures:
  * surface_pruned draws the seat quadrilateral with cv2.polylines (1 px), traces it back with
    cv2.findContours and asks cv2.pointPolygonTest(...) > -1.  Restated as: the point lies inside
    or on the quadrilateral with the truncated-to-int32 corners (crossing parity; float64 cross
    products, two separately rounded multiplies and one subtract).  The traced outline of a
    rasterised edge can differ from the ideal edge by up to half a pixel.
  * dir_pruned draws bones with cv2.line(..., 255, 5) and reads the pixel under each test point.
    Restated as: a pixel is painted iff its centre is within thickness/2 of the segment between
    the truncated end points (integer arithmetic, exact).  OpenCV's fixed-point polygon fill and
    round caps can differ on pixels at the rim of the band.
Everything else (which bones are drawn or tested, point ranges, truncations, zeroing) follows the
source line by line.
"""
import numpy as np

from .keypoints import SKELETON_MAP, dense_skeleton_2d, get_additional_coords_keypoint_5

IMG_SIZE = (224, 224)                                    # global_variables.py:75
LEG_RANGE = {                                            # global_variables.py:73-74
    "chair": list(range(40)), "swivelchair": list(range(40)),
    "bed": list(range(40)) + list(range(110, 140)), "table": list(range(40)),
    "diningtable": list(range(40)), "sofa": list(range(40)) + list(range(110, 120)),
    "car": list(range(40)) + list(range(70, 100)) + list(range(130, 160)),
}
_PASCAL_EXTRA = {"chair": 4, "swivelchair": 4, "diningtable": 4, "bed": 7, "sofa": 4}   # utils.py:253-270


def get_additional_coords_pascal(img_coords, class_name, dim=3):
    """utils.py:253-270: zero columns where keypoint-5 has midpoints."""
    n = _PASCAL_EXTRA.get(class_name)
    if n is None:
        return img_coords
    return np.concatenate([img_coords, np.zeros((dim, n))], 1)


def point_in_polygon(px, py, poly):
    """Inside or on the closed polygon `poly` (V,2) int; px, py python floats (float64)."""
    v = len(poly)
    parity = False
    for e in range(v):
        ax, ay = int(poly[e][0]), int(poly[e][1])
        bx, by = int(poly[(e + 1) % v][0]), int(poly[(e + 1) % v][1])
        cross = np.float64(bx - ax) * (np.float64(py) - np.float64(ay)) \
            - np.float64(by - ay) * (np.float64(px) - np.float64(ax))
        if cross == 0 and min(ax, bx) <= px <= max(ax, bx) and min(ay, by) <= py <= max(ay, by):
            return True
        if (ay > py) != (by > py):
            if (by > ay and cross > 0) or (by < ay and cross < 0):
                parity = not parity
    return parity


def draw_band(src, p0, p1, thickness):
    """Paint 255 on every pixel of `src` (H,W) whose centre is within thickness/2 of segment p0-p1."""
    h, w = src.shape
    x0, y0, x1, y1 = int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1])
    dx, dy = x1 - x0, y1 - y0
    l2 = dx * dx + dy * dy
    t2 = thickness * thickness
    reach = thickness // 2 + 1
    for y in range(max(0, min(y0, y1) - reach), min(h, max(y0, y1) + reach + 1)):
        for x in range(max(0, min(x0, x1) - reach), min(w, max(x0, x1) + reach + 1)):
            ux, uy = x - x0, y - y0
            dot = ux * dx + uy * dy
            if l2 == 0 or dot <= 0:
                hit = 4 * (ux * ux + uy * uy) <= t2
            elif dot >= l2:
                hit = 4 * ((x - x1) ** 2 + (y - y1) ** 2) <= t2
            else:
                cr = ux * dy - uy * dx
                hit = 4 * cr * cr <= t2 * l2
            if hit:
                src[y, x] = 255


def surface_pruned(all_coords, img_coords, seat_info, k=range(40)):
    """utils.py:274-290 (in place)."""
    points = np.array([img_coords[:2, seat_info[c]] for c in range(4)], dtype=np.int32)
    for m in k:
        j, i = all_coords[:, m]
        if point_in_polygon(float(j), float(i), points):
            all_coords[:, m] = np.zeros((2))
    return all_coords


def _facing(img_coords, seat_info):
    center_front = (img_coords[:2, seat_info[0]] + img_coords[:2, seat_info[1]]) / 2
    center_back = (img_coords[:2, seat_info[2]] + img_coords[:2, seat_info[3]]) / 2
    return (center_front - center_back)[0]


def dir_pruned(all_coords, img_coords, skeleton, seat_info, skele_info):
    """utils.py:292-325 (in place)."""
    d = _facing(img_coords, seat_info)
    src = np.zeros(IMG_SIZE, np.uint8)
    n = len(skeleton)
    if d > 0:
        prune_skeleton = [skeleton[i] for i in range(n) if "L" in skele_info[i]]
        test_skeleton = [skeleton[i] for i in range(n) if "R" in skele_info[i] or "C" in skele_info[i]]
    elif d < 0:
        prune_skeleton = [skeleton[i] for i in range(n) if "R" in skele_info[i]]
        test_skeleton = [skeleton[i] for i in range(n) if "L" in skele_info[i] or "C" in skele_info[i]]
    else:
        prune_skeleton, test_skeleton = [], []
    for (i, j) in prune_skeleton:
        draw_band(src, img_coords[:2, i].astype(int), img_coords[:2, j].astype(int), 5)
    for bone in test_skeleton:
        for m in range(8):
            test_pt_ind = skeleton.index(bone) * 10 + 1 + m
            test_pt = all_coords[:, test_pt_ind].astype(int)
            if src[test_pt[1], test_pt[0]] == 255:
                all_coords[:, test_pt_ind] = np.zeros((2))
    return all_coords


def surface_dir_prune(all_coords, img_coords, skeleton, skele_info, seat_info):
    """utils.py:327-348 (in place)."""
    d = _facing(img_coords, seat_info)
    n = len(skeleton)
    if d > 0:
        prune_skeleton = [skeleton[i] for i in range(n) if "RR" in skele_info[i]]
    elif d < 0:
        prune_skeleton = [skeleton[i] for i in range(n) if "LL" in skele_info[i]]
    else:
        prune_skeleton = []
    for bone in prune_skeleton:
        for m in range(10):
            all_coords[:, skeleton.index(bone) * 10 + m] = np.zeros((2))
    return all_coords


def make_dense_keypoints_image(sparse_keypoints, class_name, dataset="keypoint-5"):
    """make_dense_pointset.py:43-99 for one image: sparse keypoints (K,3) [x, y, present] ->
    (2, 10*n_bones) float64, or None where the reference saves nothing (no bone present)."""
    skeleton, skele_info, seat_info = SKELETON_MAP[class_name]
    img_coords = np.clip(np.asarray(sparse_keypoints), 0, 223).astype(int).T
    if dataset == "keypoint-5":
        img_coords = get_additional_coords_keypoint_5(img_coords, class_name)
        act = list(skele_info)
    else:
        img_coords = get_additional_coords_pascal(img_coords, class_name)
        truth_val = img_coords[2, :]
        act = [skele_info[i] if truth_val[b[0]] == 1 and truth_val[b[1]] == 1 else "O"
               for i, b in enumerate(skeleton)]
    if all(tag == "O" for tag in act):
        return None
    cols = []
    for (i, j), tag in zip(skeleton, act):
        cols.append(np.zeros((2, 10)) if tag == "O" else
                    np.stack([np.linspace(img_coords[d, i], img_coords[d, j], 10) for d in range(2)]))
    all_coords = np.concatenate(cols, 1)
    all_coords = surface_pruned(all_coords, img_coords, seat_info, LEG_RANGE[class_name])
    all_coords = dir_pruned(all_coords, img_coords, skeleton, seat_info, skele_info)
    if class_name in ("bed", "sofa"):
        all_coords = surface_dir_prune(all_coords, img_coords, skeleton, skele_info, seat_info)
    return all_coords
