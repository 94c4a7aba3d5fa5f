"""File-name codec, anchor projection and point depth splat of the classifier side.

Drop-in for the hot-path part of classifier_code/train_test_scripts/utils/io_utils.py:
get_params_from_file :20-31, get_render_array :97-117, get_pt_depth_map :119-124,
get_depth_map_name :126-130, get_final_file_name :132-138, name_version :147-148,
get_render_array_no_prune :164-172.  The twin parser data_creator/utils.py:359-374 is
exported as get_params_from_syn_file / img_to_depth_file.
"""
import os

import numpy as np
import torch

from . import _abi, _device as dv, pt2render_b, run_proj


def get_params_from_file(file_name):
    """'<synset>_<md5>_a<az>_e<el>_t<tilt>_d<rho>_<top>_<bottom>_<left>_<right>_.jpg'; '@' is the
    decimal point and the tilt changes sign on the way in (io_utils.py:20-31)."""
    fields = os.path.basename(file_name).split("_")
    az, el, tilt, rho = (float(f[1:].replace("@", ".")) for f in fields[2:6])
    return {"model_type": fields[0], "model_md5": fields[1], "azimuth": az, "elevation": el,
            "tilt": -tilt, "rho": rho, "crop_params": [int(f) for f in fields[6:10]]}


get_params_from_syn_file = get_params_from_file           # data_creator/utils.py:359-370


def img_to_depth_file(img):
    """data_creator/utils.py:372-374: drop the four crop fields and the extension."""
    return "_".join(img.split("_")[:-5]) + ".png1.exr"


def name_version(value):
    """io_utils.py:147-148."""
    return ("%3.3f" % value).replace(".", "@")


def _pose_stem(d):
    return "{}_{}_a{}_e{}_t{}_d{}".format(d["model_type"], d["model_md5"], name_version(d["azimuth"]),
                                          name_version(d["elevation"]), name_version(-d["tilt"]),
                                          name_version(d["rho"]))


def get_depth_map_name(file_dict, location=""):
    """io_utils.py:126-130."""
    return os.path.join(location, _pose_stem(file_dict) + ".png1.exr")


def get_final_file_name(file_dict, location=""):
    """io_utils.py:132-138."""
    c = file_dict["crop_params"]
    name = _pose_stem(file_dict) + "_{}_{}_{}_{}_.jpg".format(c[0], c[1], c[2], c[3])
    return os.path.join(location, file_dict["model_type"], file_dict["model_md5"], name)


def get_render_array_no_prune(file_dict, model_pts):
    """io_utils.py:164-172."""
    return run_proj.get_3d_to_2d(model_pts, [file_dict["azimuth"], file_dict["elevation"],
                                             file_dict["tilt"], file_dict["rho"]])


def get_render_array(file_dict, model_pts, depth_location):
    """io_utils.py:97-117 -- project, variant-B depth test, crop test + remap.

    depth_location: directory holding the depth images (file name from get_depth_map_name), or
    the (H,W) float32 depth array itself.  Returns (global_truth (N,), crop_pt_array (3,N))."""
    if isinstance(depth_location, str):
        depth = run_proj.get_depthmap_from_exr(get_depth_map_name(file_dict, depth_location))
    else:
        depth = depth_location
    cam = run_proj.camera_from_params([file_dict["azimuth"], file_dict["elevation"],
                                       file_dict["tilt"], file_dict["rho"]])
    truth, pts = run_proj.project_prune(model_pts, cam[None], depth, file_dict["crop_params"], variant="B")
    return truth[0], pts[0]


def get_pt_depth_map(render_array, size=(540, 960), radius=2):
    """io_utils.py:119-124 -- z of every point written to [v-r, v+r) x [u-r, u+r) in point
    order; the last point covering a pixel wins.  Returns (H,W) float64."""
    as_tensor = dv.is_tensor(render_array)
    pr = dv.to_dev(render_array, torch.float64)
    n = pr.shape[1]
    h, w = int(size[0]), int(size[1])
    owner = dv.empty((h * w,), torch.int32)
    out = dv.empty((h, w), torch.float64)
    _abi.check(_abi.lib.ispa_point_depth_splat(dv.ptr(pr), n, h, w, int(radius), dv.ptr(owner), dv.ptr(out),
                                               dv.stream_ptr()))
    return dv.back(out, as_tensor)
