"""Weight containers the hot path reads.

  * ZoeDepth: a state dict with HF names (``Intel/zoedepth-nyu-kitti`` safetensors, or a torch .pt/.pth
    holding the same names).  The upstream ``ZoeD_M12_NK.pt`` naming is not mapped yet.
  * CyclePose: the reference's checkpoint container written by ModelIO.save_pose_model
    (BodySLAM_not_refactored/UTILS/io_utils.py:207-232): a dict with ``model_state_dict`` (+ epoch, ate, ...).
    ``skip_linear.*`` is read from it when present (quirk Q1, cyclepose.py).
No network access is attempted: the reference's torch.hub download (interface.py:43-46) is replaced by a
local path, given explicitly or through BODYSLAM_ZOEDEPTH_WEIGHTS / BODYSLAM_CYCLEPOSE_WEIGHTS.
"""
from __future__ import annotations

import os
from typing import Dict

import torch

ENV_ZOE = "BODYSLAM_ZOEDEPTH_WEIGHTS"
ENV_POSE = "BODYSLAM_CYCLEPOSE_WEIGHTS"


def load_zoedepth_weights(path: str | None = None) -> Dict[str, torch.Tensor]:
    path = path or os.environ.get(ENV_ZOE)
    if not path:
        raise FileNotFoundError(
            f"no ZoeDepth weights: pass a path or set {ENV_ZOE} to an Intel/zoedepth-nyu-kitti state dict "
            "(.safetensors or .pt with HF names); the reference's torch.hub download is not available offline")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return sd


def load_cyclepose_checkpoint(path: str | None = None) -> Dict[str, torch.Tensor]:
    path = path or os.environ.get(ENV_POSE)
    if not path:
        raise FileNotFoundError(f"no CyclePose checkpoint: pass a path or set {ENV_POSE}")
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}
