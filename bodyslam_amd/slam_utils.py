"""The reference's 3DM helper names, importable as ``from bodyslam_amd.slam_utils import ...``
(BodySLAM_not_refactored/3DM/slam_utils.py:71-122, scaling_system.py:72-77)."""
import numpy as np

from .posegraph import PoseGraph, update_global_extrinsic  # noqa: F401
from .geom3d import (REF_DEPTH_SCALE, REF_DEPTH_TRUNC, REF_INTRINSICS, add_pose_to_list,  # noqa: F401
                     compute_curr_estimate_global_pose, ensure_so3_v2, pixel_to_3d)


def save_poses_as_kitti(poses_list, output_path) -> None:
    """Drop-in for TXTIO.save_poses_as_kitti (BodySLAM_not_refactored/UTILS/io_utils.py:264-278): one line per 4x4 pose, its
    first three rows flattened (12 numbers) and formatted with ``str`` -- byte-identical to the reference's file for equal
    inputs.  (The reference copies each pose and leaves the SO(3) correction commented out; nothing is corrected here either.)"""
    with open(output_path, "w") as f:
        for pose in poses_list:
            flat = np.asarray(pose).flatten()[:-4]
            f.write(" ".join(map(str, flat)) + "\n")
