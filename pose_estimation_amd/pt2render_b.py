"""Classifier-side flavour of the projection module.

Drop-in for classifier_code/train_test_scripts/utils/pt2render.py, which differs from the
data-creation copy only in depth_visibility_array: z <= min(3x3 depth neighbourhood + 0.05),
and the returned array is pruned to the visible points (lines 43-53).
"""
from . import pt2render as _a
from .pt2render import (camera_matrices, crop_visibility_array, ex_camera_matrix, get_camera_matrix,  # noqa: F401
                        get_render_location, rot_three_angles)


def depth_visibility_array(pt_render, depth_map, error_relaxation=0.05):
    return _a.depth_visibility_array(pt_render, depth_map, error_relaxation, variant="B")
