"""Z-buffer of the projected points themselves (an added mode: the reference reads a rendered
depth image instead, run_proj.py:33-46).  Per pixel the minimum of (float32 depth bits << 32 |
point index); `visibility` then keeps the points within `relax` of the buffer."""
import numpy as np
import torch

from . import _abi, _device as dv


def zbuffer(pt_render, size=(540, 960), radius=0, want=("winner", "depth")):
    """pt_render (3,N) or (n_pose,3,N) -> dict with 'packed' (u64 as int64 tensor/array),
    'winner' int32 (-1 empty) and 'depth' float32 (+inf empty), each (n_pose,)H,W."""
    as_tensor = dv.is_tensor(pt_render)
    pr = dv.to_dev(pt_render, torch.float64)
    single = pr.dim() == 2
    if single:
        pr = pr[None]
    n_pose, _, n = pr.shape
    h, w = int(size[0]), int(size[1])
    packed = dv.empty((n_pose, h, w), torch.int64)
    winner = dv.empty((n_pose, h, w), torch.int32) if "winner" in want else None
    depth = dv.empty((n_pose, h, w), torch.float32) if "depth" in want else None
    _abi.check(_abi.lib.ispa_zbuffer(dv.ptr(pr), n, n_pose, h, w, int(radius), dv.ptr(packed), dv.ptr(winner),
                                     dv.ptr(depth), dv.stream_ptr()))
    out = {"packed": packed, "winner": winner, "depth": depth}
    out = {k: (v[0] if single else v) for k, v in out.items() if v is not None}
    return {k: dv.back(v, as_tensor) for k, v in out.items()}


def project_zbuffer(model_pts, cams, size=(540, 960), radius=0, relax=0.0, want=("packed", "truth"), pt_render=None,
                    point_order=None):
    """Projection + z-buffer + visibility in one screen-tiled pass (ispa_project_zbuffer).

    point_order: a disparity.PointOrder of model_pts.  The pass then runs on the points in that order (what makes
    it fast, see below) and the results come back in the order of model_pts: 'truth' columns and 'winner' indices
    refer to model_pts.  ('packed' keys keep the processing-order index in their low half, and two points of
    exactly equal float32 depth on one pixel are told apart by processing order, not by their index in model_pts.)

    model_pts (N,3), cams (3,4) or (n_pose,3,4); or pt_render=(n_pose,3,N) already projected coordinates
    (ispa_zbuffer_tiled).  `want` picks the outputs: 'packed' u64 keys (as int64), 'winner' int32 (-1 empty),
    'depth' float32 (+inf empty), each (n_pose,H,W); 'truth' (n_pose,N) bool = takes part and lies within
    `relax` behind the nearest point of its pixel.  Same values as zbuffer() + visibility() on
    get_render_location's output.  Put the model's points in run_proj.spatial_order first: results are the
    same for any order, but a tile only skips the chunks of points that project elsewhere when consecutive
    points are neighbours in space."""
    from_render = pt_render is not None
    if point_order is not None and (from_render or point_order.n != len(model_pts)):
        raise ValueError("point_order goes with model_pts and must describe them")
    src_in = pt_render if from_render else model_pts
    as_tensor = dv.is_tensor(src_in)
    src = point_order.points if point_order is not None else dv.to_dev(src_in, torch.float64)
    if from_render:
        single = src.dim() == 2
        if single:
            src = src[None]
        n_pose, _, n = src.shape
        cams_t = None
    else:
        cams_t = dv.to_dev(cams, torch.float64)
        single = cams_t.dim() == 2
        cams_t = cams_t.reshape(-1, 3, 4)
        n_pose, n = cams_t.shape[0], src.shape[0]
        if src.dim() != 2 or src.shape[1] != 3:
            raise ValueError("model_pts must be (N,3)")
    h, w = int(size[0]), int(size[1])
    unknown = set(want) - {"packed", "winner", "depth", "truth"}
    if unknown or not want:
        raise ValueError(f"want must name some of packed / winner / depth / truth (got {sorted(unknown)})")
    packed = dv.empty((n_pose, h, w), torch.int64) if "packed" in want else None
    winner = dv.empty((n_pose, h, w), torch.int32) if "winner" in want else None
    depth = dv.empty((n_pose, h, w), torch.float32) if "depth" in want else None
    truth = dv.empty((n_pose, n), torch.uint8) if "truth" in want else None
    need = _abi.lib.ispa_project_zbuffer_workspace_bytes(n, n_pose, h, w)
    ws = dv.empty((need,), torch.uint8)
    if from_render:
        _abi.check(_abi.lib.ispa_zbuffer_tiled(dv.ptr(src), n, n_pose, h, w, int(radius), float(relax), dv.ptr(packed),
                                               dv.ptr(winner), dv.ptr(depth), dv.ptr(truth), dv.ptr(ws), need,
                                               dv.stream_ptr()))
    else:
        _abi.check(_abi.lib.ispa_project_zbuffer(dv.ptr(src), n, dv.ptr(cams_t), n_pose, h, w, int(radius), float(relax),
                                                 dv.ptr(packed), dv.ptr(winner), dv.ptr(depth), dv.ptr(truth),
                                                 dv.ptr(ws), need, dv.stream_ptr()))
    if point_order is not None:                      # back to the order of model_pts
        if truth is not None:
            truth = truth[:, point_order.inverse.long()]
        if winner is not None:
            rank = torch.cat([point_order.rank, point_order.rank.new_full((1,), -1)])     # index -1 -> -1
            winner = rank[winner.long()]
    out = {"packed": packed, "winner": winner, "depth": depth, "truth": None if truth is None else truth.bool()}
    out = {k: (v[0] if single else v) for k, v in out.items() if v is not None}
    return {k: dv.back(v, as_tensor) for k, v in out.items()}


def _check_images(d, n_pose):
    """The kernels index image p of a (n_pose,H,W) stack for pose p: anything else would read out of bounds."""
    if d.dim() != 3 or d.shape[0] != n_pose:
        raise ValueError(f"need one (H,W) image per pose: pt_render has {n_pose} pose(s), the image stack has shape "
                         f"{tuple(d.shape)}")


def visibility(pt_render, depth, relax=0.0):
    """visible = lands in the image with finite positive z, and z - depth[y,x] <= relax.

    `depth` is the float32 depth image, or the 'packed' int64 image of zbuffer() itself (then no
    winner / depth decode is needed in between)."""
    as_tensor = dv.is_tensor(pt_render)
    pr = dv.to_dev(pt_render, torch.float64)
    is_packed = (depth.dtype in (torch.int64, torch.uint64)) if dv.is_tensor(depth) else np.asarray(depth).dtype.kind in "iu"
    if is_packed:
        d = dv.to_dev(depth if dv.is_tensor(depth) else np.asarray(depth).view(np.int64), torch.int64)
        single = pr.dim() == 2
        if single:
            pr, d = pr[None], d[None]
        n_pose, _, n = pr.shape
        _check_images(d, n_pose)
        h, w = d.shape[-2:]
        truth = dv.empty((n_pose, n), torch.uint8)
        _abi.check(_abi.lib.ispa_zbuffer_visibility_packed(dv.ptr(pr), n, n_pose, dv.ptr(d), h, w, float(relax),
                                                           dv.ptr(truth), dv.stream_ptr()))
        truth = truth.bool()
        return dv.back(truth[0] if single else truth, as_tensor)
    d = dv.to_dev(depth, torch.float32)
    single = pr.dim() == 2
    if single:
        pr, d = pr[None], d[None]
    n_pose, _, n = pr.shape
    _check_images(d, n_pose)
    h, w = d.shape[-2:]
    truth = dv.empty((n_pose, n), torch.uint8)
    _abi.check(_abi.lib.ispa_zbuffer_visibility(dv.ptr(pr), n, n_pose, dv.ptr(d), h, w, float(relax),
                                                dv.ptr(truth), dv.stream_ptr()))
    truth = truth.bool()
    return dv.back(truth[0] if single else truth, as_tensor)
