"""ctypes binding of libispa_hip.so (declarations mirror include/ispa_abi.h).

There is no CPU fallback: if the shared library is missing or does not export a symbol,
importing this module raises.
"""
import ctypes
import os

# torch before the library: libispa_hip.so and torch must share ONE HIP runtime (device pointers and streams pass
# between them), and the dynamic loader keeps whichever libamdhip64.so.7 comes first.  With the library first
# the system copy wins and torch's own bundled stack is left half replaced: the first HIP call then fails with
# "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISPA_HIP_LIBRARY points at another build of the same library (development builds)
LIB_PATH = os.environ.get("ISPA_HIP_LIBRARY") or os.path.join(_HERE, "libispa_hip.so")

ISPA_ABI_VERSION = 1
VIS_A, VIS_B = 0, 1
F32, BF16, F32_SPLIT, F32_SPLIT_ALWAYS = 0, 1, 2, 3
NN_NORMALIZE = 256
NN_NO_PAD = 512
PCK_MAX_RADII = 8
PRUNE_POLYGON, PRUNE_BAND, PRUNE_DROP = 1, 2, 4
PRUNE_MAX_VERTICES, PRUNE_MAX_SEGMENTS = 64, 256

_p = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double
_sz = ctypes.c_size_t

# name -> (restype, argtypes); one entry per declaration in include/ispa_abi.h
SIGNATURES = {
    "ispa_abi_version": (_i, []),
    "ispa_last_error": (ctypes.c_char_p, []),
    "ispa_project": (_i, [_p, _i64, _p, _i, _p, _p]),
    "ispa_depth_visibility": (_i, [_p, _i64, _i, _p, _i, _i, _i, _i, _i, _d, _i, _p, _p]),
    "ispa_crop_remap": (_i, [_p, _i64, _i, _p, _i, _i, _p, _p]),
    "ispa_project_prune": (_i, [_p, _i64, _p, _i, _p, _i, _i, _i, _i, _i, _d, _i, _p, _i, _p, _p, _p, _p, _p]),
    "ispa_flow_splat": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p, _p, _p]),
    "ispa_flow_splat_hw": (_i, [_p, _p, _p, _i64, _i, _i, _i, _i, _p, _p, _p]),
    "ispa_make_disparity": (_i, [_p, _i64, _p, _p, _i, _p, _i, _i, _i, _i, _i, _d, _p, _i, _i, _p, _p]),
    "ispa_make_disparity_workspace_bytes": (_sz, [_i64, _i]),
    "ispa_make_disparity_ws": (_i, [_p, _i64, _p, _p, _i, _p, _i, _i, _i, _i, _i, _d, _p, _i, _i, _p, _p, _sz, _p]),
    "ispa_make_disparity_ordered": (_i, [_p, _p, _p, _i64, _p, _p, _i, _p, _i, _i, _i, _i, _i, _d, _p, _i, _i, _p, _p, _sz,
                                         _p]),
    "ispa_point_depth_splat": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p]),
    "ispa_zbuffer": (_i, [_p, _i64, _i, _i, _i, _i, _p, _p, _p, _p]),
    "ispa_zbuffer_visibility": (_i, [_p, _i64, _i, _p, _i, _i, _d, _p, _p]),
    "ispa_zbuffer_visibility_packed": (_i, [_p, _i64, _i, _p, _i, _i, _d, _p, _p]),
    "ispa_project_zbuffer_workspace_bytes": (_sz, [_i64, _i, _i, _i]),
    "ispa_project_zbuffer": (_i, [_p, _i64, _p, _i, _i, _i, _i, _d, _p, _p, _p, _p, _p, _sz, _p]),
    "ispa_zbuffer_tiled": (_i, [_p, _i64, _i, _i, _i, _i, _d, _p, _p, _p, _p, _p, _sz, _p]),
    "ispa_nn_workspace_bytes": (_sz, [_i, _i, _i64, _i64, _i]),
    "ispa_nn_workspace_bytes_for": (_sz, [_i, _i, _i, _i64, _i64, _i]),
    "ispa_nn_topk": (_i, [_p, _p, _i, _i, _i, _i64, _i64, _i, _i, _p, _p, _p, _sz, _p, _p]),
    "ispa_nn_noalloc_scratch_bytes": (_sz, []),
    "ispa_nn_topk_noalloc": (_i, [_p, _p, _i, _i, _i, _i64, _i64, _i, _i, _p, _p, _p, _p]),
    "ispa_timing_enable": (_i, [_i]),
    "ispa_timing_collect": (_i, [_p, _i, _p]),
    "ispa_l2norm": (_i, [_p, _i, _i, _i64, _p, _p]),
    "ispa_resize_bilinear": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "ispa_corr_map_eccv": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ispa_corr_map_eccv_views": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ispa_corr_map_acmm_workspace_floats": (_sz, [_i, _i, _i]),
    "ispa_corr_map_acmm": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ispa_extract_feature": (_i, [_p, _i, _i, _i, _i, _p, _i64, _i, _i, ctypes.c_float, _i, _p, _p]),
    "ispa_nearest_view": (_i, [_p, _i, _p, _i, _p, _p, _p]),
    "ispa_pck_counts": (_i, [_p, _i, _i, _i64, _p, _i, _d, _i, _p, _i, _p, _p]),
    "ispa_hard_negatives": (_i, [_p, _i, _i, _i64, _p, _i, _d, _i, _d, _p, _p, _p]),
    "ispa_prune_points_2d": (_i, [_p, _i, _i64, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p, _p]),
}


class IspaError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C pose_estimation_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "pose_estimation_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.ispa_abi_version()
    if got != ISPA_ABI_VERSION:
        raise ImportError(f"libispa_hip.so has ABI version {got}, binding expects {ISPA_ABI_VERSION}")
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise IspaError(f"ispa error {rc}: {lib.ispa_last_error().decode(errors='replace')}")
