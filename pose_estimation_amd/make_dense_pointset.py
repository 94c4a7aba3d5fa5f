"""`python -m pose_estimation_amd.make_dense_pointset --dataset keypoint-5 --class_name chair`

Script entry point of data_creator/make_dense_pointset.py:11-115 for annotated real images: per
image the sparse keypoints (sparse_keypoints/<stem>.npy, (K,3) rows [x, y, present]) become 10
points per skeleton bone, pruned by seat outline / far-side bones / facing direction, saved as
dense_keypoints/<stem>.npy = (2, 10 * n_bones) float64, and the stems are listed in
image_lists/dense_keypoints/<dataset>_<class>_<subset>.txt.  All images of a directory go through
the pruning kernel in batches; the image files themselves are not read (the reference only prints
their shape)."""
import argparse
import os

import numpy as np

from . import dense_pointset, global_variables as gv


def setup(dataset, class_name, subset="pascal"):
    """make_dense_pointset.py:11-26 (directory layout)."""
    data_dir = os.path.join(gv.processed_image_loc, dataset)
    stem = class_name + "_" + subset if dataset == "pascal3D" else class_name
    image_loc = os.path.join(data_dir, stem, "images")
    sparse_keypoint_loc = os.path.join(data_dir, stem, "sparse_keypoints")
    dense_keypoint_loc = os.path.join(data_dir, stem, "dense_keypoints")
    return data_dir, image_loc, os.listdir(image_loc), sparse_keypoint_loc, dense_keypoint_loc


def make_dense_keypoints(dataset, class_name, subset, batch=1024):
    _, _, img_list, sparse_keypoint_loc, dense_keypoint_loc = setup(dataset, class_name, subset)
    os.makedirs(dense_keypoint_loc, exist_ok=True)
    list_dir = os.path.join(gv.data_dir, "image_lists", "dense_keypoints")
    os.makedirs(list_dir, exist_ok=True)
    written = 0
    with open(os.path.join(list_dir, "_".join([dataset, class_name, subset]) + ".txt"), "w") as info_file:
        for start in range(0, len(img_list), batch):
            stems = [name.split(".")[0] for name in img_list[start:start + batch]]
            sparse = np.stack([np.load(os.path.join(sparse_keypoint_loc, stem + ".npy")) for stem in stems])
            all_coords, saved = dense_pointset.make_dense_keypoints(sparse, class_name, dataset)
            for stem, coords, keep in zip(stems, all_coords, saved):
                if keep:
                    info_file.write(stem + "\n")
                    np.save(os.path.join(dense_keypoint_loc, stem + ".npy"), coords)
                    written += 1
    return written


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset", default="keypoint-5", help="The dataset for processing.")
    parser.add_argument("--class_name", default="chair", help="The dataset the class to processed")
    parser.add_argument("--subset", default="pascal", help="Only for Pascal Dataset.")
    args = parser.parse_args()
    print(make_dense_keypoints(args.dataset, args.class_name, args.subset), "dense keypoint files written")


if __name__ == "__main__":
    main()
