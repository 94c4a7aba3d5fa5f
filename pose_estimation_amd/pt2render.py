"""Point projection / visibility entry points with the reference's names and array layouts.

Drop-in for data_creator/render_wt_pt_proj/pt2render.py (visibility variant A, relaxation
0.15).  The classifier-side copy of that module, which tests z <= min(3x3 depth) + 0.05 and
returns a pruned array, is `pose_estimation_amd.pt2render_b`.

Arrays may be numpy (results come back as numpy, in-place contracts are honoured on the
caller's array) or torch tensors on the HIP device (results stay on the device).  The camera
matrix builders are host numpy code: they produce twelve numbers per pose.
"""
import numpy as np
import torch

from . import _abi, _device as dv
from .variables import VIS_XLIM, VIS_YLIM

__all__ = ["get_render_location", "crop_visibility_array", "depth_visibility_array",
           "rot_three_angles", "ex_camera_matrix", "get_camera_matrix", "camera_matrices"]


# ---- camera (host) ---------------------------------------------------------------------------
def _axis_rotations(yaw, pitch, roll):
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    about_z = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    about_y = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    about_x = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return about_z, about_y, about_x


def rot_three_angles(yaw, pitch, roll):
    """pt2render.py:58-73 -- degrees in, (3,3) out; elevation sign flipped, transposed, rows
    rotated up by one to match the renderer's axes."""
    rz, ry, rx = _axis_rotations(np.deg2rad(yaw), np.deg2rad(-pitch), np.deg2rad(roll))
    world_to_cam = np.matmul(rz, np.matmul(ry, rx)).T
    return np.roll(world_to_cam, shift=-1, axis=0)


def ex_camera_matrix(yaw, pitch, roll, rho):
    """pt2render.py:76-84 -- (3,4) extrinsic [R | t], camera at distance rho on the optical axis."""
    ext = np.zeros((3, 4))
    ext[:, :3] = rot_three_angles(yaw, pitch, roll)
    ext[:, 3] = -np.array([0, 0, rho])
    ext[1:, :] = -ext[1:, :]
    return ext


def get_camera_matrix(intrinsic, extrinsic, world_matrix):
    """pt2render.py:86-87."""
    return np.matmul(np.matmul(intrinsic, extrinsic), world_matrix)


def camera_matrices(poses, intrinsic=None, world_matrix=None):
    """Batched P = K E W for poses (n,4) = [azimuth, elevation, tilt, rho] -> (n,3,4) float64.

    Vectorised closed form of the three functions above for pose sweeps (thousands of poses);
    agrees with the per-pose functions to a few ulp.
    """
    from .variables import INTRINSIC_CAMERA_MATRIX, WORLD_MATRIX
    k = INTRINSIC_CAMERA_MATRIX if intrinsic is None else intrinsic
    w = WORLD_MATRIX if world_matrix is None else world_matrix
    poses = np.atleast_2d(np.asarray(poses, dtype=np.float64))
    a, p, r = np.deg2rad(poses[:, 0]), np.deg2rad(-poses[:, 1]), np.deg2rad(poses[:, 2])
    n = len(poses)
    rz = np.zeros((n, 3, 3)); ry = np.zeros((n, 3, 3)); rx = np.zeros((n, 3, 3))
    rz[:, 0, 0] = np.cos(a); rz[:, 0, 1] = -np.sin(a); rz[:, 1, 0] = np.sin(a); rz[:, 1, 1] = np.cos(a); rz[:, 2, 2] = 1
    ry[:, 0, 0] = np.cos(p); ry[:, 0, 2] = np.sin(p); ry[:, 1, 1] = 1; ry[:, 2, 0] = -np.sin(p); ry[:, 2, 2] = np.cos(p)
    rx[:, 0, 0] = 1; rx[:, 1, 1] = np.cos(r); rx[:, 1, 2] = -np.sin(r); rx[:, 2, 1] = np.sin(r); rx[:, 2, 2] = np.cos(r)
    m = np.matmul(rz, np.matmul(ry, rx)).transpose(0, 2, 1)
    ext = np.zeros((n, 3, 4))
    ext[:, :, :3] = np.roll(m, -1, axis=1)
    ext[:, 2, 3] = -poses[:, 3]
    ext[:, 1:, :] = -ext[:, 1:, :]
    return np.matmul(np.matmul(np.asarray(k, dtype=np.float64), ext), np.asarray(w, dtype=np.float64))


# ---- device entry points -----------------------------------------------------------------------
def get_render_location(pt_array, camera_matrix):
    """pt2render.py:6-13 -- (N,3) points, (3,4) camera -> (3,N) rows [round(u), round(v), z].

    A (n_pose,3,4) stack of cameras returns (n_pose,3,N)."""
    as_tensor = dv.is_tensor(pt_array)
    pts = dv.to_dev(pt_array, torch.float64)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError(f"pt_array must be (N,3), got {tuple(pts.shape)}")
    cams = dv.to_dev(camera_matrix, torch.float64)
    batched = cams.dim() == 3
    cams = cams.reshape(-1, 3, 4)
    n, n_pose = pts.shape[0], cams.shape[0]
    out = dv.empty((n_pose, 3, n), torch.float64)
    _abi.check(_abi.lib.ispa_project(dv.ptr(pts), n, dv.ptr(cams), n_pose, dv.ptr(out), dv.stream_ptr()))
    if not batched:
        out = out[0]
    return dv.back(out, as_tensor)


def depth_visibility_array(pt_render, depth_map, error_relaxation=0.15, variant="A"):
    """pt2render.py:43-55 -- truth[i] = pixel gate and |z_i - depth[y_i, x_i]| <= relaxation.

    Returns (truth, pt_render); pt_render is returned unchanged (variant A does not prune)."""
    as_tensor = dv.is_tensor(pt_render)
    pr = dv.to_dev(pt_render, torch.float64)
    if pr.dim() != 2 or pr.shape[0] != 3:
        raise ValueError(f"pt_render must be (3,N), got {tuple(pr.shape)}")
    depth = dv.to_dev(depth_map, torch.float32)
    if depth.dim() != 2:
        raise ValueError("depth_map must be (H,W)")
    h, w = depth.shape
    n = pr.shape[1]
    truth = dv.empty((n,), torch.uint8)
    code = _abi.VIS_A if variant == "A" else _abi.VIS_B
    _abi.check(_abi.lib.ispa_depth_visibility(dv.ptr(pr), n, 1, dv.ptr(depth), 0, h, w, min(VIS_YLIM, h),
                                              min(VIS_XLIM, w), float(error_relaxation), code,
                                              dv.ptr(truth), dv.stream_ptr()))
    truth = truth.bool()
    if variant == "A":
        return dv.back(truth, as_tensor), pt_render
    if as_tensor:
        return truth, pr[:, truth]
    t = truth.cpu().numpy()
    return t, pt_render[:, t]


def crop_visibility_array(pt_render, crop_params, resize=True):
    """pt2render.py:18-38 -- strict crop test on the render coordinates, then rows 0 and 1 of
    the CALLER'S array are rewritten in place in 224-pixel crop coordinates.

    crop_params = [top, bottom, left, right].  Returns (truth, pt_render)."""
    as_tensor = dv.is_tensor(pt_render)
    crop = torch.tensor([int(c) for c in crop_params], dtype=torch.int32, device=dv.device())
    if as_tensor:
        if not (pt_render.is_cuda and pt_render.dtype == torch.float64 and pt_render.is_contiguous()):
            raise ValueError("in-place crop needs a contiguous float64 device tensor")
        pr = pt_render
    else:
        if not (isinstance(pt_render, np.ndarray) and pt_render.dtype == np.float64):
            raise ValueError("in-place crop needs a float64 ndarray")
        pr = dv.to_dev(pt_render, torch.float64)
    if pr.dim() != 2 or pr.shape[0] != 3:
        raise ValueError(f"pt_render must be (3,N), got {tuple(pr.shape)}")
    n = pr.shape[1]
    truth = dv.empty((n,), torch.uint8)
    _abi.check(_abi.lib.ispa_crop_remap(dv.ptr(pr), n, 1, dv.ptr(crop), 0, 1 if resize else 0,
                                        dv.ptr(truth), dv.stream_ptr()))
    if as_tensor:
        return truth.bool(), pt_render
    pt_render[:2, :] = pr[:2].cpu().numpy()
    return truth.bool().cpu().numpy(), pt_render
