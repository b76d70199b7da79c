"""CPU oracle for the 3DM hot path (pose chain + back-projection) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

numpy restatement of
  * compute_curr_estimate_global_pose   BodySLAM_not_refactored/3DM/slam_utils.py:110-122
  * ensure_so3_v2                       BodySLAM_not_refactored/3DM/slam_utils.py:93-108
                                        (dup. UTILS/geometry_utils.py:137-153)
  * add_pose_to_list(invert_matrix)     BodySLAM_not_refactored/3DM/slam_utils.py:71-85
  * the sequence loop that chains them  BodySLAM_not_refactored/EVALUATION/MPEM_eval.py:216-223,
                                        3DM/slam.py:148-153
  * pixel_to_3d                         BodySLAM_not_refactored/3DM/scaling_system.py:72-77
  * the RGBD depth constants (u16 / depth_scale, >= depth_trunc zeroed; valid iff z > 0)
                                        BodySLAM_not_refactored/3DM/slam_utils.py:173,212-220,232
                                        3DM/slam.py:25-29 (intrinsics, depth_scale = 1000)

Pinned by tests/test_oracle_geom3d.py against tests/golden/geom3d_*.npz, produced by
oracle/make_golden.py from the reference's own functions imported in the build container.
The Open3D calls on this path (create_from_depth_image, tsdf.integrate) are third-party C++ not
under /root/reference: for those "parity unpinned"; the in-repo formula above is what we follow.
The integer part (validity mask, row-major compaction indices) also exists as plain C in
oracle/backproject_ref.c.
"""
from __future__ import annotations

import numpy as np

# slam.py:25-29
REF_INTRINSICS = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)
REF_DEPTH_SCALE = 1000.0
REF_DEPTH_TRUNC = 3.0


def ensure_so3_v2(matrix: np.ndarray) -> np.ndarray:
    U, _, Vt = np.linalg.svd(matrix)
    D = np.eye(3)
    D[2, 2] = np.linalg.det(U) * np.linalg.det(Vt)
    return np.dot(U, np.dot(D, Vt))


def compute_curr_estimate_global_pose(global_extrinsic: np.ndarray, transformation: np.ndarray) -> np.ndarray:
    g = np.dot(global_extrinsic, transformation)
    g[:3, :3] = ensure_so3_v2(g[:3, :3])
    return g


def pose_chain(t_rel: np.ndarray, g0: np.ndarray | None = None) -> np.ndarray:
    """t_rel float32 [N,4,4] -> absolute float64 [N+1,4,4] starting at g0 (identity)."""
    g = np.eye(4) if g0 is None else np.array(g0, dtype=np.float64)
    out = [g.copy()]
    for t in t_rel:
        g = compute_curr_estimate_global_pose(g, t)
        out.append(g.copy())
    return np.stack(out)


def pixel_to_3d(u, v, depth, fx, fy, cx, cy):
    x = (u - cx) * depth / fx
    y = (v - cy) * depth / fy
    z = depth
    return np.array([x, y, z])


def backproject(depth_u16: np.ndarray, K=REF_INTRINSICS, depth_scale=REF_DEPTH_SCALE,
                depth_trunc=REF_DEPTH_TRUNC, pose: np.ndarray | None = None):
    """depth_u16 [H,W] -> (xyz float32 [M,3], idx int32 [M]) for valid pixels in row-major order.

    z = d / depth_scale (float32 division, as Open3D's convert/RGBD path and numpy on a float32
    image do), zeroed where z >= depth_trunc; valid iff z > 0.  x, y follow pixel_to_3d evaluated
    in float64 then rounded once to float32; with ``pose`` (4x4 float64, camera->world) the point
    is transformed in float64 before that rounding.
    """
    fx, fy, cx, cy = K
    H, W = depth_u16.shape
    z32 = depth_u16.astype(np.float32) / np.float32(depth_scale)
    z32 = np.where(z32 >= np.float32(depth_trunc), np.float32(0), z32)
    valid = z32 > 0
    idx = np.flatnonzero(valid.ravel()).astype(np.int32)
    v, u = np.divmod(idx.astype(np.int64), W)
    z = z32.ravel()[idx].astype(np.float64)
    x = (u - cx) * z / fx
    y = (v - cy) * z / fy
    pts = np.stack([x, y, z], axis=1)
    if pose is not None:
        pts = pts @ pose[:3, :3].T + pose[:3, 3]
    return pts.astype(np.float32), idx
