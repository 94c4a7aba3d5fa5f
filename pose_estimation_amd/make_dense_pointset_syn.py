"""`python -m pose_estimation_amd.make_dense_pointset_syn --class_name chair`

Script entry point of data_creator/make_dense_pointset_syn.py:16-79: anchors -> skeleton ->
10 points per bone, projected, pruned (depth variant A + crop) per render of the template model,
saved as dense_keypoints/<image stem>.npy = (2, 10 * n_bones) float64."""
import argparse
import os

import numpy as np

from . import dense_pointset, global_variables as gv, io_utils, run_proj


def make_dense_keypoints(class_name, batch=256):
    template_model, _ = gv.syn_template[class_name]
    anchors = np.loadtxt(os.path.join(gv.anchor_loc, template_model + ".txt"), delimiter=" ")
    save_loc = gv.syn_dense_keypoint_loc
    os.makedirs(save_loc, exist_ok=True)
    images = sorted(x for x in os.listdir(gv.g_save_file_location_final) if template_model in x)
    for start in range(0, len(images), batch):
        chunk = images[start:start + batch]
        dicts = [io_utils.get_params_from_syn_file(x) for x in chunk]
        params = [[d["azimuth"], d["elevation"], d["tilt"], d["rho"]] for d in dicts]
        crops = [d["crop_params"] for d in dicts]
        depth = run_proj.load_depth_stack(os.path.join(gv.depth_loc, io_utils.img_to_depth_file(x)) for x in chunk)
        keys = dense_pointset.dense_keypoints_for_renders(anchors, class_name, params, depth, crops)
        for name, key_2d in zip(chunk, keys):
            np.save(os.path.join(save_loc, name.split(".")[0]), key_2d)
    return len(images)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--class_name", default="chair", help="The dataset the class to processed")
    args = parser.parse_args()
    print(make_dense_keypoints(args.class_name), "dense keypoint files written to", gv.syn_dense_keypoint_loc)


if __name__ == "__main__":
    main()
