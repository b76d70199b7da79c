"""3DM on MI355X: SE(3) pose chain with per-step SO(3) projection and depth -> point-cloud
back-projection, plus the reference's scalar helper names as drop-ins.

Mirrors (same names, argument meaning and return types):
  compute_curr_estimate_global_pose, ensure_so3_v2, add_pose_to_list
                                    BodySLAM_not_refactored/3DM/slam_utils.py:71-122
  pixel_to_3d                       BodySLAM_not_refactored/3DM/scaling_system.py:72-77
Batched siblings (what a sequence actually calls): pose_chain, backproject.
All arithmetic runs in the HIP library; there is no CPU fallback.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

# BodySLAM_not_refactored/3DM/slam.py:25-29, slam_utils.py:173
REF_INTRINSICS = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)
REF_DEPTH_SCALE = 1000.0
REF_DEPTH_TRUNC = 3.0


def _dev(device: int = 0) -> torch.device:
    L.init(device)
    return torch.device("cuda", device)


def pose_chain(t_rel, g0=None, device: int = 0) -> torch.Tensor:
    """relative poses float32 [N,4,4] (torch GPU tensor or numpy) -> absolute poses float64 [N+1,4,4]
    on the GPU; element 0 is g0 (identity; a numpy (4,4) or a GPU tensor).  G_i = ensure_so3(G_{i-1} @ T_i)
    (slam_utils.py:110-122)."""
    dev = _dev(device)
    t = torch.as_tensor(t_rel, dtype=torch.float32).to(dev).reshape(-1, 16).contiguous()
    N = t.shape[0]
    out = torch.empty(N + 1, 16, dtype=torch.float64, device=dev)
    if isinstance(g0, torch.Tensor) and g0.is_cuda:      # continue a chain from a pose already on the device
        L.pose_chain_from(t, N, g0.to(torch.float64).reshape(16).contiguous(), out)
    else:
        L.pose_chain(t, N, None if g0 is None else np.asarray(g0, dtype=np.float64).reshape(16), out)
    return out.view(N + 1, 4, 4)


def backproject(depth_u16: torch.Tensor, K: Sequence[float] = REF_INTRINSICS, depth_scale: float = REF_DEPTH_SCALE,
                depth_trunc: float = REF_DEPTH_TRUNC, poses: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """depth uint16 payload [B,H,W] (torch int16/uint16 GPU tensor) -> (xyz fp32 [B,H*W,3], idx int32 [B,H*W],
    count int32 [B]); the first count[b] entries of image b are its valid pixels in row-major order.
    poses: optional float64 [B,4,4] camera->world."""
    assert depth_u16.is_cuda and depth_u16.dim() == 3 and depth_u16.element_size() == 2
    B, H, W = depth_u16.shape
    L.init(depth_u16.device.index or 0)
    dev = depth_u16.device
    xyz = torch.empty(B, H * W, 3, device=dev)
    idx = torch.empty(B, H * W, dtype=torch.int32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(1, B * (H * W // 256 + 2)), dtype=torch.int32, device=dev)
    pz = None
    if poses is not None:
        pz = poses.to(dev, dtype=torch.float64).reshape(B, 16).contiguous()
    L.backproject(depth_u16.contiguous(), K, depth_scale, depth_trunc, pz, xyz, idx, cnt, scratch, B, H, W)
    return xyz, idx, cnt


# ---- the reference's own names -------------------------------------------------------------------
def ensure_so3_v2(matrix: np.ndarray) -> np.ndarray:
    """Closest rotation to a 3x3 matrix: U diag(1, 1, det(U) det(V^T)) V^T (slam_utils.py:93-108)."""
    m = np.asarray(matrix, dtype=np.float64)
    if m.shape != (3, 3):
        raise ValueError(f"Invalid rotation matrix shape {m.shape}.")
    g0 = np.eye(4)
    g0[:3, :3] = m
    # one chain step with the identity motion projects the rotation block
    out = pose_chain(np.eye(4, dtype=np.float32)[None], g0=g0)
    return out[1, :3, :3].cpu().numpy()


def compute_curr_estimate_global_pose(global_extrinsic: np.ndarray, transformation: np.ndarray) -> np.ndarray:
    """(4,4) float64 x (4,4) float32 -> (4,4) float64 (slam_utils.py:110-122)."""
    t = np.asarray(transformation)
    if t.shape != (4, 4) or np.asarray(global_extrinsic).shape != (4, 4):
        raise ValueError(f"Invalid motion matrix shape {t.shape}.")
    return pose_chain(t.astype(np.float32)[None], g0=np.asarray(global_extrinsic, dtype=np.float64))[1].cpu().numpy()


def add_pose_to_list(matrix, pose_list: List[np.ndarray], invert_matrix: bool = False):
    """slam_utils.py:71-85 (bookkeeping: the 4x4 inverse is host numpy, as in the reference)."""
    if not isinstance(matrix, np.ndarray):
        matrix = matrix.cpu().numpy()
    if invert_matrix:
        matrix = np.linalg.inv(matrix)
    pose_list.append(matrix)


def pixel_to_3d(u, v, depth, fx, fy, cx, cy) -> np.ndarray:
    """scaling_system.py:72-77."""
    dev = _dev(0)
    uvd = torch.tensor([[float(u), float(v), float(depth)]], dtype=torch.float64, device=dev)
    out = torch.empty(1, 3, dtype=torch.float64, device=dev)
    L.pixel_to_3d(uvd, (fx, fy, cx, cy), out, 1)
    return out[0].cpu().numpy()
