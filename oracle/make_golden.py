"""Generate tests/golden/*.npz by running the REFERENCE's own code (and HF ZoeDepth, the installed
weight-compatible restatement of the un-vendored upstream network) in the build container.

    python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]

This script is the only place that imports /root/reference; nothing from it is copied into the
repo and it is never run on the GPU box.  The fixtures hold inputs (or the seeds that generate
them), and the reference's outputs.

Reference entry points exercised (paths relative to /root/reference/BodySLAM_not_refactored):
  3DM/slam_utils.py:110-122   compute_curr_estimate_global_pose      -> geom3d_chain.npz
  3DM/slam_utils.py:93-108    ensure_so3_v2                          -> geom3d_chain.npz (so3_*)
  3DM/slam_utils.py:71-85     add_pose_to_list(invert_matrix=True)   -> geom3d_chain.npz (inv_*)
  3DM/scaling_system.py:72-77 pixel_to_3d                            -> geom3d_backproject.npz
  MPEM/architecture_v3.py:108-239 ConditionalGenerator(mode="pose")  -> cyclepose_pose.npz
  UTILS/geometry_utils.py:230-265 quaternion_to_matrix / normalize   -> cyclepose_pose.npz (quat_*)
HF (transformers 5.15.0) ZoeDepthForDepthEstimation, ZoeD_NK config  -> zoedepth_tiny.npz,
                                                                         zoedepth_full.npz
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import cyclepose_ref as CP  # noqa: E402  (weights generator + shapes only)
from oracle import zoedepth_ref as Z    # noqa: E402  (weights generator + config only)


def random_rel_poses(n: int, seed: int) -> np.ndarray:
    """float32 [n,4,4]: small random motions built the way MPEM builds them (t | wxyz quaternion ->
    matrix in float32), so the rotation blocks are only float32-orthonormal."""
    rng = np.random.default_rng(seed)
    q = np.concatenate([np.ones((n, 1)), 0.05 * rng.standard_normal((n, 3))], axis=1).astype(np.float32)
    t = (0.01 * rng.standard_normal((n, 3))).astype(np.float32)
    q = q / np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    r, i, j, k = q.T
    two_s = np.float32(2.0) / (q * q).sum(-1)
    R = np.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                  two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                  two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    T[:, :3, :3] = R.reshape(n, 3, 3).astype(np.float32)
    T[:, :3, 3] = t
    return T


def synth_depth_u16(h: int, w: int, seed: int) -> np.ndarray:
    """uint16 depth with all three classes: 0 (invalid), 1..2999 (valid), >= 3000 (truncated)."""
    rng = np.random.default_rng(seed)
    d = rng.integers(200, 3400, size=(h, w)).astype(np.uint16)
    d[rng.random((h, w)) < 0.15] = 0
    d[0, 0] = 2999
    d[0, 1] = 3000
    d[-1, -1] = 1
    d[-1, -2] = 65535
    return d


def gen_geom3d(ref: str, out: str):
    for m in ("open3d", "open3d.core", "open3d.visualization", "cv2", "filterpy", "filterpy.kalman",
              "torchvision", "torchvision.transforms", "tsdf"):
        sys.modules.setdefault(m, MagicMock())
    sys.path.insert(0, os.path.join(ref, "BodySLAM_not_refactored", "3DM"))
    sys.path.insert(0, os.path.join(ref, "BodySLAM_not_refactored"))
    import slam_utils            # reference
    import scaling_system        # reference

    t_rel = random_rel_poses(1000, seed=7)
    g = np.eye(4)
    g_abs, inv_list = [g.copy()], []
    slam_utils.add_pose_to_list(g, inv_list, invert_matrix=True)
    for t in t_rel:
        g = slam_utils.compute_curr_estimate_global_pose(g, t)
        g_abs.append(g.copy())
        slam_utils.add_pose_to_list(g, inv_list, invert_matrix=True)
    rng = np.random.default_rng(3)
    so3_in = rng.standard_normal((16, 3, 3))
    so3_in[0] = np.diag([1.0, 1.0, -1.0])            # reflection: exercises the det correction
    so3_out = np.stack([slam_utils.ensure_so3_v2(m) for m in so3_in])
    np.savez_compressed(os.path.join(out, "geom3d_chain.npz"), t_rel=t_rel, g_abs=np.stack(g_abs),
                        inv_abs=np.stack(inv_list), so3_in=so3_in, so3_out=so3_out)

    K = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)   # slam.py:25-28
    d = synth_depth_u16(48, 64, seed=11)
    # reference semantics: depth/1000 as float32 image, >= 3.0 zeroed (slam_utils.py:173,212-220), valid z > 0
    z32 = d.astype(np.float32) / np.float32(1000.0)
    z32[z32 >= np.float32(3.0)] = 0
    idx, xyz = [], []
    for v in range(d.shape[0]):
        for u in range(d.shape[1]):
            if z32[v, u] > 0:
                idx.append(v * d.shape[1] + u)
                xyz.append(scaling_system.pixel_to_3d(u, v, float(z32[v, u]), *K))
    np.savez_compressed(os.path.join(out, "geom3d_backproject.npz"), depth=d, K=np.array(K),
                        idx=np.array(idx, dtype=np.int32), xyz=np.array(xyz, dtype=np.float64))

    # KITTI pose text written by the reference's own writer (UTILS/io_utils.py:264-278) for the first 24 absolute poses
    from UTILS import io_utils   # reference
    io_utils.TXTIO().save_poses_as_kitti(g_abs[:24], os.path.join(out, "kitti_poses_24.txt"))


def gen_cyclepose(ref: str, out: str):
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, os.path.join(ref, "BodySLAM_not_refactored"))
    import MPEM.architecture_v3 as A  # reference
    gen = A.ConditionalGenerator(input_shape=(6, 256, 256), device="cpu").eval()
    # quirk Q1 (architecture_v3.py:208-209): skip_linear is created lazily; create it up front so
    # the explicit weight is the one used.
    gen.skip_linear = torch.nn.Linear(CP.SKIP_FEATURES, 7)
    w = CP.synth_weights(seed=5)
    missing, unexpected = gen.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(3, 6, 128, 128)).astype(np.float32))
    with torch.no_grad():
        T = gen(x, mode="pose")
        # intermediates through the reference's own sub-modules
        c0 = gen.initial_model(x)
        c2 = gen.downsampling(c0)
        pooled = gen.pose_conv(c2).flatten(1)
    q = torch.from_numpy(rng.standard_normal((8, 4)).astype(np.float32))
    qn = A.PO.normalize_quaternion(q)
    Rm = A.PO.quaternion_to_matrix(qn)
    np.savez_compressed(os.path.join(out, "cyclepose_pose.npz"), weight_seed=5, input_seed=21,
                        T=T.numpy(), c0_mean=c0.mean(dim=(2, 3)).numpy(), c0_sample=c0[:, :, ::16, ::16].numpy(),
                        c2_sample=c2[:, ::8, ::4, ::4].numpy(), pooled=pooled.numpy(),
                        quat_in=q.numpy(), quat_norm=qn.numpy(), quat_R=Rm.numpy())


def hf_config(c: Z.ZoeConfig):
    from transformers import ZoeDepthConfig
    backbone_config = dict(model_type="beit", image_size=c.image_size, num_hidden_layers=c.layers,
                           hidden_size=c.hidden, intermediate_size=c.intermediate, num_attention_heads=c.heads,
                           use_relative_position_bias=True, reshape_hidden_states=False,
                           out_features=[f"stage{t}" for t in c.taps])
    return ZoeDepthConfig(
        backbone_config=backbone_config, neck_hidden_sizes=list(c.neck_hidden), fusion_hidden_size=c.fusion,
        reassemble_factors=list(c.reassemble_factors), readout_type="project", add_projection=c.add_projection,
        num_relative_features=c.rel_features, bottleneck_features=c.bottleneck, bin_embedding_dim=c.bin_dim,
        num_attractors=[16, 8, 4, 1], attractor_alpha=1000, attractor_gamma=2, attractor_kind="mean",
        min_temp=c.min_temp, max_temp=c.max_temp, bin_centers_type="softplus",
        bin_configurations=[{"name": n_, "n_bins": 64, "min_depth": 1e-3, "max_depth": {"nyu": 10.0, "kitti": 80.0}[n_]}
                            for n_ in c.head_names],
        num_patch_transformer_layers=c.pt_layers, patch_transformer_hidden_size=c.pt_hidden,
        patch_transformer_intermediate_size=c.pt_inter, patch_transformer_num_attention_heads=c.pt_heads)


def gen_zoedepth(out: str, full: bool):
    from transformers import ZoeDepthForDepthEstimation
    # tiny backbone, full-size neck + heads; both routes forced in turn
    cfg = Z.tiny_config()
    res = {}
    for tag, rb in (("nyu", 3.0), ("kitti", -3.0)):
        w = Z.synth_weights(cfg, seed=1, route_bias=rb)
        m = ZoeDepthForDepthEstimation(hf_config(cfg)).eval()
        m.load_state_dict(w, strict=True)
        rng = np.random.default_rng(31)
        x = torch.from_numpy(rng.standard_normal((1, 3, 64, 96), dtype=np.float32))
        with torch.no_grad():
            o = m(pixel_values=x)
        res[f"depth_{tag}"] = o.predicted_depth.numpy()
        res[f"logits_{tag}"] = o.domain_logits.numpy()
    np.savez_compressed(os.path.join(out, "zoedepth_tiny.npz"), weight_seed=1, input_seed=31, **res)
    if full:
        cfg = Z.ZOED_NK
        w = Z.synth_weights(cfg, seed=1)
        m = ZoeDepthForDepthEstimation(hf_config(cfg)).eval()
        m.load_state_dict(w, strict=True)
        rng = np.random.default_rng(32)
        x = torch.from_numpy(rng.standard_normal((1, 3, 384, 512), dtype=np.float32))
        with torch.no_grad():
            o = m(pixel_values=x)
        d = o.predicted_depth.numpy()
        np.savez_compressed(os.path.join(out, "zoedepth_full.npz"), weight_seed=1, input_seed=32,
                            depth_sub=d[:, ::8, ::8], depth_mean=d.mean(), depth_absmean=np.abs(d).mean(),
                            logits=o.domain_logits.numpy())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--skip-full", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    gen_geom3d(a.ref, a.out)
    gen_cyclepose(a.ref, a.out)
    for m in [k for k, v in sys.modules.items() if isinstance(v, MagicMock) or k == "cv2"]:
        del sys.modules[m]      # the stubs must not leak into the transformers import below
    gen_zoedepth(a.out, full=not a.skip_full)
    print("golden vectors written to", a.out)
