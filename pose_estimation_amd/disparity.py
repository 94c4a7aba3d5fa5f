"""Proxy optical flow ("disparity") maps between a render and the same model turned by (10, 10).

Drop-in for data_creator/make_disparity_syn.py: `make_disparity_maps` is the data-in /
data-out core of the per-image body (lines 44-68), batched over images; `make_disparity`
keeps the script's directory-walking entry point (lines 17-70).  `demo_flow` is the flow demo
of data_creator/render_wt_pt_proj/get_of.py:38-62 (render resolution, no crop, both poses'
truths).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _abi, _device as dv, io_utils, pt2render, run_proj

FLOW_SIDE = 224          # make_disparity_syn.py:61
SPREAD = 2               # make_disparity_syn.py:63
POSE_STEP = (10.0, 10.0)  # make_disparity_syn.py:49-50
DEMO_SIZE = (540, 960)   # get_of.py:54
DEMO_STEP = (10.0, 0.0)  # get_of.py:43-44


def flow_splat(uv1, uv2, truth, side=FLOW_SIDE, spread=SPREAD):
    """make_disparity_syn.py:55-68 for already projected points.

    uv1, uv2 (n_map,3,N) or (3,N) crop coordinates at pose 1 / pose 2, truth (n_map,N) = truth_1.
    side: an int (square maps) or (H, W) as in get_of.py:54.
    Returns flow (n_map,H,W,2) float64 (or (H,W,2))."""
    as_tensor = dv.is_tensor(uv1)
    a = dv.to_dev(uv1, torch.float64)
    b = dv.to_dev(uv2, torch.float64)
    t = dv.to_dev(truth, torch.uint8)
    single = a.dim() == 2
    if single:
        a, b, t = a[None], b[None], t[None]
    n_map, _, n = a.shape
    h, w = (int(side), int(side)) if np.isscalar(side) else (int(side[0]), int(side[1]))
    # square maps of up to 8 x 25088 pixels (side <= 448) run the banded LDS kernel, which needs no owner image
    owner = None if h == w and h * w <= 8 * 25088 else dv.empty((n_map, h * w), torch.int32)
    flow = dv.empty((n_map, h, w, 2), torch.float64)
    _abi.check(_abi.lib.ispa_flow_splat_hw(dv.ptr(a), dv.ptr(b), dv.ptr(t), n, n_map, int(spread), h, w,
                                           dv.ptr(owner), dv.ptr(flow), dv.stream_ptr()))
    if single:
        flow = flow[0]
    return dv.back(flow, as_tensor)


class PointOrder:
    """A model's points in a processing order of its own (by default along a Hilbert curve, run_proj.spatial_order),
    with what the kernels need to keep the reference's order as the one that decides which point wins a pixel:
    `points` (N,3) device tensor in processing order, `rank` (N) int32 = each point's index in the original array,
    `inverse` (N) int32 = where each original point went.  Build it once per model and pass it to
    make_disparity_maps(point_order=...): same flow maps bit for bit, neighbouring lanes gather neighbouring depth
    texels."""

    def __init__(self, key_3d, perm=None):
        pts = key_3d.detach().cpu().numpy() if dv.is_tensor(key_3d) else np.asarray(key_3d, dtype=np.float64)
        if perm is None:                             # non-finite coordinates only need SOME place in the order
            perm = run_proj.spatial_order(np.nan_to_num(pts, nan=0.0, posinf=0.0, neginf=0.0))
        perm = np.asarray(perm, dtype=np.int64)
        if perm.shape != (pts.shape[0],) or not np.array_equal(np.sort(perm), np.arange(pts.shape[0])):
            raise ValueError("perm must be a permutation of the model's point indices")
        inverse = np.empty_like(perm)
        inverse[perm] = np.arange(perm.shape[0])
        self.n = int(pts.shape[0])
        self.points = dv.to_dev(np.ascontiguousarray(pts[perm]), torch.float64)
        self.rank = dv.to_dev(perm.astype(np.int32), torch.int32)
        self.inverse = dv.to_dev(inverse.astype(np.int32), torch.int32)


def make_disparity_maps(key_3d, params, depth_maps, crop_params, fused=True, point_order=None, spread=SPREAD):
    """Body of make_disparity for a batch of renders of ONE model.

    key_3d (N,3) surface samples; params (n_map,4) = [azimuth, elevation, tilt, rho] per render;
    depth_maps (n_map,540,960) f32 (the first pose's depth image, used for both poses as the
    reference does); crop_params (n_map,4).  Only the first pose's truth masks the points.
    fused=True runs the whole body in one to three launches without materialising projected coordinates
    (ispa_make_disparity_ws; ispa_make_disparity is the one-launch form for callers without a workspace); fused=False goes through
    project_prune twice and flow_splat, materialising the projected coordinates like the reference.
    point_order: a PointOrder of key_3d (fused path only): the points are processed in its order, the results
    are those of the given order.  spread: half the side of the block a point writes (2 in the script, line 63).
    Returns flow (n_map,224,224,2) float64, numpy unless key_3d is a device tensor."""
    as_tensor = dv.is_tensor(key_3d)
    if point_order is not None and (not fused or point_order.n != len(key_3d)):
        raise ValueError("point_order goes with the fused path and must describe key_3d")
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    n_map = params.shape[0]
    cams1 = run_proj.cameras_from_params(params)
    more = params.copy()
    more[:, 0] += POSE_STEP[0]
    more[:, 1] += POSE_STEP[1]
    cams2 = run_proj.cameras_from_params(more)
    pts = dv.to_dev(key_3d, torch.float64)
    depth = dv.to_dev(depth_maps, torch.float32)
    if depth.dim() == 2:
        depth = depth[None]
    crops = dv.rows(crop_params, 4, torch.int32)
    if crops.shape[0] != n_map:
        raise ValueError("need one crop box per render")
    if fused:
        h, w = depth.shape[-2:]
        if depth.shape[0] not in (1, n_map):
            raise ValueError("need one depth map per render or a single shared one")
        c1 = dv.to_dev(cams1, torch.float64)
        c2 = dv.to_dev(cams2, torch.float64)
        crop_t = crops
        flow = dv.empty((n_map, FLOW_SIDE, FLOW_SIDE, 2), torch.float64)
        # with a workspace the per-point work runs once per (render, point), not once per band
        need = _abi.lib.ispa_make_disparity_workspace_bytes(pts.shape[0], n_map)
        ws = dv.empty((max(need, 4),), torch.uint8)
        po = point_order
        _abi.check(_abi.lib.ispa_make_disparity_ordered(
            dv.ptr(po.points if po else pts), dv.ptr(po.rank) if po else None, dv.ptr(po.inverse) if po else None,
            pts.shape[0], dv.ptr(c1), dv.ptr(c2), n_map, dv.ptr(depth), 1 if depth.shape[0] > 1 else 0,
            h, w, min(run_proj.VIS_YLIM, h), min(run_proj.VIS_XLIM, w), 0.15, dv.ptr(crop_t), 1, int(spread),
            dv.ptr(flow), dv.ptr(ws), need, dv.stream_ptr()))
        return dv.back(flow, as_tensor)
    if depth.shape[0] not in (1, n_map):
        raise ValueError("need one depth map per render or a single shared one")
    shared = depth[0] if depth.shape[0] == 1 else depth       # a 2-D map is shared by all poses
    truth1, uv1 = run_proj.project_prune(pts, cams1, shared, crops)
    _, uv2 = run_proj.project_prune(pts, cams2, shared, crops)
    flow = flow_splat(uv1, uv2, truth1, spread=int(spread))
    return dv.back(flow, as_tensor)


def demo_flow(key_3d, params, depth_maps, step=DEMO_STEP, spread=SPREAD):
    """data_creator/render_wt_pt_proj/get_of.py:38-62 for a batch of views of ONE model: project at `params` and at
    the pose turned by `step` (azimuth +10 in the script), prune both against the SAME depth image without a crop,
    keep the points visible at both poses (line 48) and splat (uv2 - uv1) of the first pose's render coordinates
    into a flow map of the depth image's size.

    key_3d (N,3); params (n_map,4) or [azimuth, elevation, tilt, rho] (the script truncates them to integers,
    line 29: pass them truncated to reproduce it); depth_maps (n_map,H,W) f32, or one (H,W) image shared by all
    views.  Returns flow (n_map,H,W,2) float64 ((H,W,2) for a single params row), numpy unless key_3d is a
    device tensor."""
    as_tensor = dv.is_tensor(key_3d)
    single = np.ndim(params) == 1
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    more = params.copy()
    more[:, 0] += step[0]
    more[:, 1] += step[1]
    cams1 = run_proj.cameras_from_params(params)
    cams2 = run_proj.cameras_from_params(more)
    pts = dv.to_dev(key_3d, torch.float64)
    depth = dv.to_dev(depth_maps, torch.float32)
    if depth.dim() == 3 and depth.shape[0] == 1:
        depth = depth[0]
    if depth.dim() == 3 and depth.shape[0] != params.shape[0]:
        raise ValueError("need one depth map per view or a single shared one")
    truth1, uv1 = run_proj.project_prune(pts, cams1, depth)
    truth2, uv2 = run_proj.project_prune(pts, cams2, depth)
    flow = flow_splat(uv1, uv2, truth1 & truth2, side=tuple(depth.shape[-2:]), spread=spread)
    if single:
        flow = flow[0]
    return dv.back(flow, as_tensor)


def make_disparity(class_name, key_3d, image_dir, depth_dir, save_dir, template_model, batch=64, io_threads=8):
    """make_disparity_syn.py:17-70 with the data locations passed in instead of read from
    global_variables: for every render of `template_model` under image_dir, write
    <save_dir>/<image stem>.npy = (224,224,2) float64 flow.  Returns the number of maps written.

    The files either side of the GPU call move on `io_threads` host threads: the depth images of the next batch
    are decoded while the current one is on the GPU, and the .npy files are written behind it (the script reads,
    computes and writes one render at a time).  File names and contents are those of the script."""
    del class_name
    os.makedirs(save_dir, exist_ok=True)
    images = sorted(x for x in os.listdir(image_dir) if template_model in x)
    order = PointOrder(key_3d)                       # once per model; every batch of renders re-uses it
    chunks = [images[start:start + batch] for start in range(0, len(images), batch)]

    def read_depth(name):
        return run_proj.get_depthmap_from_exr(os.path.join(depth_dir, io_utils.img_to_depth_file(name)))

    with ThreadPoolExecutor(max_workers=max(1, int(io_threads))) as pool:
        reads = [pool.submit(read_depth, x) for x in chunks[0]] if chunks else []
        writes = []
        for n, chunk in enumerate(chunks):
            depth = np.stack([f.result() for f in reads])
            reads = [pool.submit(read_depth, x) for x in chunks[n + 1]] if n + 1 < len(chunks) else []
            dicts = [io_utils.get_params_from_syn_file(x) for x in chunk]
            params = [[d["azimuth"], d["elevation"], d["tilt"], d["rho"]] for d in dicts]
            crops = [d["crop_params"] for d in dicts]
            flows = make_disparity_maps(key_3d, params, depth, crops, point_order=order)
            writes += [pool.submit(np.save, os.path.join(save_dir, name.split(".")[0]), flow)
                       for name, flow in zip(chunk, flows)]
        for f in writes:
            f.result()                               # re-raises what a writer thread hit
    return len(images)
