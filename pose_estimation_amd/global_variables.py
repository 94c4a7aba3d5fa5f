"""Data locations and model tables of the data-creation scripts.

Same names as data_creator/global_variables.py:6-21,77-83 so that the script entry points find
their inputs where the reference's do; every location can be redirected by assigning to the
module attribute (or through the ISPA_DATA_DIR environment variable for the root)."""
import os

BASE_DIR = os.getcwd()
data_dir = os.environ.get("ISPA_DATA_DIR") or os.path.join(BASE_DIR, "../data")
synthetic_data_dir = os.path.join(data_dir, "synthetic_data")
anchor_loc = os.path.join(synthetic_data_dir, "anchors")
depth_loc = os.path.join(synthetic_data_dir, "depth")
syn_disparity_loc = os.path.join(synthetic_data_dir, "disparity")
syn_dense_keypoint_loc = os.path.join(synthetic_data_dir, "dense_keypoints")
g_save_file_location_final = os.path.join(synthetic_data_dir, "render_final")
processed_image_loc = os.path.join(data_dir, "real_data")
sample_points_loc = os.path.join(data_dir, "sample_points")
g_datasets_folder = os.path.abspath(os.path.join(data_dir, "datasets"))
g_shapenet_root_folder = os.path.join(g_datasets_folder, "ShapeNetCore/ShapeNetCore.v1")

# class -> (template model id, ShapeNet synset)
syn_template = {
    "chair": ("13fdf00cde077f562f6f52615fb75fca", "03001627"),
    "swivelchair": ("1e92f53f3a191323d42b9650f19dd425", "03001627"),
    "sofa": ("1fd45c57ab27cb6cea65c47b660136e7", "04256520"),
    "bed": ("7c8eb4ab1f2c8bfa2fb46fb8b9b1ac9f", "02818832"),
    "table": ("e41da371550711697062f2d72cde5c95", "04379243"),
    "car": ("1aeee7288b89ec1cc805dfe4ca9f2fdb", "02958343"),
}
