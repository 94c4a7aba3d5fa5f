"""Camera constants of the renderer (data_creator/render_wt_pt_proj/variables.py:3-9)."""
import numpy as np

# world (x, y, z) -> (x, -z, y)
WORLD_MATRIX = np.array([[1, 0, 0, 0],
                         [0, 0, -1, 0],
                         [0, 1, 0, 0],
                         [0, 0, 0, 1]])
# focal 1050 px, principal point (480, 270): a 960 x 540 render
INTRINSIC_CAMERA_MATRIX = np.array([[1050, 0, 480],
                                    [0, 1050, 270],
                                    [0, 0, 1]])
RENDER_H, RENDER_W = 540, 960
VIS_YLIM, VIS_XLIM = 540, 950      # gate of depth_visibility_array (pt2render.py:47)
