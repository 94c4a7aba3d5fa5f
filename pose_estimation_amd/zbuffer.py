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
    h, w = d.shape[-2:]
    truth = dv.empty((n_pose, n), torch.uint8)
    _abi.check(_abi.lib.ispa_zbuffer_visibility(dv.ptr(pr), n, n_pose, dv.ptr(d), h, w, float(relax),
                                                dv.ptr(truth), dv.stream_ptr()))
    truth = truth.bool()
    return dv.back(truth[0] if single else truth, as_tensor)
