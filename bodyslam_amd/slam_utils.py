"""The reference's 3DM helper names, importable as ``from bodyslam_amd.slam_utils import ...``
(BodySLAM_not_refactored/3DM/slam_utils.py:71-122, scaling_system.py:72-77)."""
from .geom3d import (REF_DEPTH_SCALE, REF_DEPTH_TRUNC, REF_INTRINSICS, add_pose_to_list,  # noqa: F401
                     compute_curr_estimate_global_pose, ensure_so3_v2, pixel_to_3d)
