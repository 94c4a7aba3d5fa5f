"""`python -m pose_estimation_amd.make_disparity_syn --class_name chair`

Script entry point of data_creator/make_disparity_syn.py:17-80: for every render of the class's
template model under render_final/, write disparity/<image stem>.npy = (224,224,2) float64 flow.
The per-image body runs as one fused HIP launch per batch of renders.

The 100k surface samples come from run_proj.get_keypoints (an area-weighted sampler over the model's .obj,
no trimesh needed), or from sample_points/<synset>_<model>.npy when the model file is missing (the reference
falls back to a fixed file the same way, make_disparity_syn.py:28-32); the depth renders are decoded by
run_proj.get_depthmap_from_exr (pose_estimation_amd.exr, no OpenEXR needed)."""
import argparse
import os

import numpy as np

from . import disparity, global_variables as gv, run_proj


def get_keypoints(model_file, model_synset, template_model, k=100000):
    """run_proj.get_keypoints on the model file, else the cached samples (make_disparity_syn.py:28-32)."""
    if os.path.exists(model_file):
        return run_proj.get_keypoints(model_file, k)
    return np.load(os.path.join(gv.sample_points_loc, model_synset + "_" + template_model + ".npy"))


def make_disparity(class_name, batch=64):
    template_model, model_synset = gv.syn_template[class_name]
    model_file = os.path.join(gv.g_shapenet_root_folder, model_synset, template_model, "model.obj")
    key_3d = get_keypoints(model_file, model_synset, template_model)
    return disparity.make_disparity(class_name, key_3d, gv.g_save_file_location_final, gv.depth_loc,
                                    gv.syn_disparity_loc, template_model, batch=batch)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--class_name", default="chair", help="The dataset the class to processed")
    args = parser.parse_args()
    print(make_disparity(args.class_name), "disparity maps written to", gv.syn_disparity_loc)


if __name__ == "__main__":
    main()
