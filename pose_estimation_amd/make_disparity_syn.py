"""`python -m pose_estimation_amd.make_disparity_syn --class_name chair`

Script entry point of data_creator/make_disparity_syn.py:17-80: for every render of the class's
template model under render_final/, write disparity/<image stem>.npy = (224,224,2) float64 flow.
The per-image body runs as one fused HIP launch per batch of renders.

Inputs the reference gets from trimesh / OpenEXR are read from caches here: the 100k surface
samples from sample_points/<synset>_<model>.npy (the file the reference itself falls back to,
make_disparity_syn.py:32) unless trimesh is installed, and depth images through
run_proj.get_depthmap_from_exr (an .npy next to the .exr, or OpenEXR if installed)."""
import argparse
import os

import numpy as np

from . import disparity, global_variables as gv


def get_keypoints(model_file, model_synset, template_model, k=100000):
    """run_proj.get_keypoints (run_proj.py:15-22) when trimesh is importable, else the cached samples."""
    try:
        import trimesh
        mesh = trimesh.load(model_file)
        if type(mesh) == list:
            for j in range(1, len(mesh)):
                mesh[0] = mesh[0].union(mesh[j])
            mesh = mesh[0]
        return np.asarray(mesh.sample(int(k), 0))
    except Exception:
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
