"""MI355X-native dense correspondence / disparity stage of iSPA-Net-style pose estimation.

Python entry points keep the names and array layouts of val-iisc/pose_estimation; the work
is done by hand-written HIP kernels in libispa_hip.so (include/ispa_abi.h).  Importing the
package loads that library and fails if it is missing: there is no CPU path.
"""
from . import _abi  # noqa: F401  (loads libispa_hip.so or raises)
from . import (correspondence, dense_pointset, disparity, global_variables, io_utils, make_dense_pointset,  # noqa: F401
               make_dense_pointset_syn, make_disparity_syn, pt2render, pt2render_b, run_proj, variables, zbuffer)

__version__ = "0.1.0"
