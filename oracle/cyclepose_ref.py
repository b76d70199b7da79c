"""CPU oracle for the MPEM hot path (CyclePose generator, mode="pose") -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates, as plain functional torch fp32 over a state-dict with the reference's own parameter
names:
  * ConditionalGenerator.forward(mode="pose")  BodySLAM_not_refactored/MPEM/architecture_v3.py:195-226
      initial_model :120-125, downsampling :129-139, pose_conv :143-147, pose_dense :150-155,
      skip_linear :208-211 (lazily created in the reference -- quirk Q1; here an explicit weight)
  * PoseOperator.normalize_quaternion / quaternion_to_matrix
      BodySLAM_not_refactored/UTILS/geometry_utils.py:263-265, :230-260
  * the input transform of MPEMInterface.infer_relative_pose_between
      BodySLAM_not_refactored/MPEM/mpem_interface.py:40-44,85-94  (CenterCrop(128), ToTensor,
      Normalize(0.5, 0.5), channel concat)

Pinned by tests/test_oracle_cyclepose.py against tests/golden/cyclepose_*.npz, which
oracle/make_golden.py produced by importing the reference's own module in the build container.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SKIP_FEATURES = 512 + 256 * 32 * 32  # architecture_v3.py:205 at 128x128 input


def param_shapes() -> Dict[str, Tuple[int, ...]]:
    return {
        "initial_model.1.weight": (64, 6, 7, 7), "initial_model.1.bias": (64,),
        "downsampling.0.weight": (128, 64, 3, 3), "downsampling.0.bias": (128,),
        "downsampling.3.weight": (256, 128, 3, 3), "downsampling.3.bias": (256,),
        "pose_conv.0.weight": (512, 256, 3, 3), "pose_conv.0.bias": (512,),
        "pose_dense.1.weight": (128, 512), "pose_dense.1.bias": (128,),
        "pose_dense.3.weight": (7, 128), "pose_dense.3.bias": (7,),
        "skip_linear.weight": (7, SKIP_FEATURES), "skip_linear.bias": (7,),
    }


def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def synth_weights(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic pose-branch weights (no checkpoint ships with the reference)."""
    out = {}
    for name, shape in param_shapes().items():
        rng = np.random.default_rng(_name_seed(name, seed))
        x = rng.standard_normal(int(np.prod(shape)), dtype=np.float32).reshape(shape)
        if name.endswith("bias"):
            x = 0.1 * x
        else:
            x = x / math.sqrt(int(np.prod(shape[1:])))
        out[name] = torch.from_numpy(x)
    return out


def center_crop_pair(frames_u8: torch.Tensor, pair_idx: torch.Tensor, crop: int = 128) -> torch.Tensor:
    """uint8 [N,H,W,3] + pairs [P,2] -> float32 [P,6,crop,crop] (mpem_interface.py:40-44,85-94).
    torchvision CenterCrop offsets: top = round((H-crop)/2), left = round((W-crop)/2)."""
    N, H, W, _ = frames_u8.shape
    top = int(round((H - crop) / 2.0))
    left = int(round((W - crop) / 2.0))
    x = frames_u8[:, top:top + crop, left:left + crop, :].permute(0, 3, 1, 2).to(torch.float32) / 255.0
    x = (x - 0.5) / 0.5
    return torch.cat([x[pair_idx[:, 0]], x[pair_idx[:, 1]]], dim=1)


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def pose7(w: Dict[str, torch.Tensor], x: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    """x [P,6,128,128] -> raw pose vector [P,7] (t, q_wxyz un-normalised)."""
    y = F.pad(x, (3, 3, 3, 3), mode="reflect")
    y = F.relu(F.instance_norm(F.conv2d(y, w["initial_model.1.weight"], w["initial_model.1.bias"]), eps=1e-5))
    if taps is not None:
        taps["c0"] = y
    y = F.relu(F.instance_norm(F.conv2d(y, w["downsampling.0.weight"], w["downsampling.0.bias"], stride=2, padding=1), eps=1e-5))
    if taps is not None:
        taps["c1"] = y
    y = F.relu(F.instance_norm(F.conv2d(y, w["downsampling.3.weight"], w["downsampling.3.bias"], stride=2, padding=1), eps=1e-5))
    if taps is not None:
        taps["c2"] = y
    c = F.relu(F.conv2d(y, w["pose_conv.0.weight"], w["pose_conv.0.bias"], stride=2, padding=1))
    pooled = c.mean(dim=(2, 3))
    if taps is not None:
        taps["pooled"] = pooled
    cat = torch.cat([pooled, y.reshape(y.shape[0], -1)], dim=1)
    skip = F.linear(cat, w["skip_linear.weight"], w["skip_linear.bias"])
    dense = F.linear(F.relu(F.linear(pooled, w["pose_dense.1.weight"], w["pose_dense.1.bias"])),
                     w["pose_dense.3.weight"], w["pose_dense.3.bias"])
    return dense + skip


def pose_matrix(p7: torch.Tensor) -> torch.Tensor:
    """[P,7] -> [P,4,4] (architecture_v3.py:218-226)."""
    t = p7[:, :3]
    q = p7[:, 3:]
    q = q / torch.norm(q, p=2, dim=-1, keepdim=True)
    R = quaternion_to_matrix(q)
    T = torch.eye(4, dtype=p7.dtype).unsqueeze(0).repeat(p7.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    return T


def forward_pose(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    with torch.no_grad():
        return pose_matrix(pose7(w, x))
