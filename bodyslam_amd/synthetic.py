"""Synthetic endoscopy-like sequences for the bench and the parity tests:
per channel a sum of 8 random 2-D sinusoids (mean ~106/255, like the reference's fixture images),
i.i.d. noise sigma = 8, and a cumulative 1-px shift per frame so that consecutive frames differ."""
from __future__ import annotations

import numpy as np


def make_sequence(n: int, h: int = 480, w: int = 640, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h + n, dtype=np.float32), np.arange(w + n, dtype=np.float32), indexing="ij")
    base = np.zeros((h + n, w + n, 3), np.float32)
    for c in range(3):
        f = np.full((h + n, w + n), 106.0, np.float32)
        for _ in range(8):
            fy, fx = rng.uniform(0.002, 0.02, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(5, 20)
            f += amp * np.sin(2 * np.pi * (fy * yy + fx * xx) + ph).astype(np.float32)
        base[..., c] = f
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        fr = base[i:i + h, i:i + w] + rng.normal(0.0, 8.0, size=(h, w, 3)).astype(np.float32)
        out[i] = np.clip(np.rint(fr), 0, 255).astype(np.uint8)
    return out


def make_sequence_at(indices, total: int, h: int = 480, w: int = 640, seed: int = 0) -> np.ndarray:
    """frames `indices` of ONE synthetic sequence of `total` frames (the construction of make_sequence -- the same smooth field cropped along its
    diagonal -- with the noise of frame i drawn from its own generator, so any rank can make exactly the frames it owns): what a multi-GPU run
    needs to process ONE sequence cut into blocks instead of one unrelated sequence per rank.

    The field is never materialised (at 8 ranks x 25 steps x 128 frames it would be 26 081 x 26 241 x 3 floats per rank): a term
    amp sin(2 pi (fy y + fx x) + ph) is separable, sin(a_y + ph) cos(b_x) + cos(a_y + ph) sin(b_x), so a frame's window costs two outer products
    per term.  Frame i depends on (seed, total, i) only, never on which other frames are asked for."""
    rng = np.random.default_rng(seed)
    n = int(total)
    terms = []
    for c in range(3):
        for _ in range(8):
            fy, fx = rng.uniform(0.002, 0.02, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(5, 20)
            terms.append((c, float(fy), float(fx), float(ph), float(amp)))
    indices = [int(i) for i in indices]
    fyc, fxc, phc, ampc = (np.array([[t[j] for t in terms if t[0] == c] for c in range(3)]) for j in (1, 2, 3, 4))
    out = np.empty((len(indices), h, w, 3), np.uint8)
    for k, i in enumerate(indices):
        assert 0 <= i < n, (i, n)
        yy = np.arange(i, i + h, dtype=np.float64)[:, None]
        xx = np.arange(i, i + w, dtype=np.float64)[None, :]
        fr = np.random.default_rng([seed, i]).standard_normal(size=(h, w, 3), dtype=np.float32).astype(np.float64) * 8.0 + 106.0
        for c in range(3):          # the channel's eight terms as one [h, 16] x [16, w] product
            ay = 2 * np.pi * fyc[c][None, :] * yy + phc[c][None, :]                   # [h, 8]
            bx = 2 * np.pi * fxc[c][:, None] * xx                                       # [8, w]
            # (einsum, not `@`: a 480 x 16 x 640 product is far below where a threaded BLAS pays -- 13 ms against 94 ms on 8 cores)
            fr[..., c] += np.einsum("hk,kw->hw", np.concatenate([np.sin(ay) * ampc[c][None, :], np.cos(ay) * ampc[c][None, :]], 1),
                                    np.concatenate([np.cos(bx), np.sin(bx)], 0))
        out[k] = np.clip(np.rint(fr), 0, 255).astype(np.uint8)
    return out


# ---------------------------------------------------------------------------------------------------
# random-init weights of the two networks' architectures (no checkpoint is reachable offline)
# ---------------------------------------------------------------------------------------------------
def zoedepth_param_shapes(cfg) -> dict:
    """HF state-dict names -> shapes of a ZoeD_NK-style model (bodyslam_amd.zoedepth.ZoeConfig)."""
    s = {}
    H, I = cfg.hidden, cfg.intermediate
    s["backbone.beit.embeddings.cls_token"] = (1, 1, H)
    s["backbone.beit.embeddings.patch_embeddings.projection.weight"] = (H, 3, cfg.patch, cfg.patch)
    s["backbone.beit.embeddings.patch_embeddings.projection.bias"] = (H,)
    nrd = (2 * (cfg.image_size // cfg.patch) - 1) ** 2 + 3
    for l in range(cfg.layers):
        p = f"backbone.beit.layers.{l}."
        for n, sh in (("lambda_1", (H,)), ("lambda_2", (H,)), ("attention.q_proj.weight", (H, H)), ("attention.q_proj.bias", (H,)),
                      ("attention.k_proj.weight", (H, H)), ("attention.v_proj.weight", (H, H)), ("attention.v_proj.bias", (H,)),
                      ("attention.o_proj.weight", (H, H)), ("attention.o_proj.bias", (H,)), ("layernorm_before.weight", (H,)),
                      ("layernorm_before.bias", (H,)), ("layernorm_after.weight", (H,)), ("layernorm_after.bias", (H,)),
                      ("mlp.fc1.weight", (I, H)), ("mlp.fc1.bias", (I,)), ("mlp.fc2.weight", (H, I)), ("mlp.fc2.bias", (H,)),
                      ("relative_position_bias.relative_position_bias_table", (nrd, cfg.heads))):
            s[p + n] = sh
    for i, (c, f) in enumerate(zip(cfg.neck_hidden, (4, 2, 1, 0.5))):
        p = f"neck.reassemble_stage.layers.{i}."
        s[p + "projection.weight"], s[p + "projection.bias"] = (c, H, 1, 1), (c,)
        if f > 1:
            s[p + "resize.weight"], s[p + "resize.bias"] = (c, c, int(f), int(f)), (c,)
        elif f < 1:
            s[p + "resize.weight"], s[p + "resize.bias"] = (c, c, 3, 3), (c,)
        s[f"neck.reassemble_stage.readout_projects.{i}.0.weight"] = (H, 2 * H)
        s[f"neck.reassemble_stage.readout_projects.{i}.0.bias"] = (H,)
        s[f"neck.convs.{i}.weight"] = (cfg.fusion, c, 3, 3)
    F_ = cfg.fusion
    for i in range(4):
        p = f"neck.fusion_stage.layers.{i}."
        s[p + "projection.weight"], s[p + "projection.bias"] = (F_, F_, 1, 1), (F_,)
        for r in ("residual_layer1", "residual_layer2"):
            for c in ("convolution1", "convolution2"):
                s[p + f"{r}.{c}.weight"], s[p + f"{r}.{c}.bias"] = (F_, F_, 3, 3), (F_,)
    if getattr(cfg, "add_projection", True):
        s["relative_head.projection.weight"], s["relative_head.projection.bias"] = (256, 256, 3, 3), (256,)
    s["relative_head.conv1.weight"], s["relative_head.conv1.bias"] = (F_ // 2, F_, 3, 3), (F_ // 2,)
    s["relative_head.conv2.weight"], s["relative_head.conv2.bias"] = (cfg.rel_features, F_ // 2, 3, 3), (cfg.rel_features,)
    s["relative_head.conv3.weight"], s["relative_head.conv3.bias"] = (1, cfg.rel_features, 1, 1), (1,)
    B, E, D = cfg.bottleneck, cfg.bin_dim, cfg.pt_hidden
    s["metric_head.conv2.weight"], s["metric_head.conv2.bias"] = (B, B, 1, 1), (B,)
    single = len(cfg.head_names) == 1            # ZoeD_N / ZoeD_K: HF ZoeDepthMetricDepthEstimationHead (modeling_zoedepth.py:1106-1200)
    SM, PM, HID = cfg.seed_mlp, cfg.proj_mlp, cfg.clb_hidden
    if not single:
        for l in range(cfg.pt_layers):
            p = f"metric_head.patch_transformer.transformer_encoder.{l}."
            for n in ("query", "key", "value", "out_proj"):
                s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (D, D), (D,)
            s[p + "linear1.weight"], s[p + "linear1.bias"] = (cfg.pt_inter, D), (cfg.pt_inter,)
            s[p + "linear2.weight"], s[p + "linear2.bias"] = (D, cfg.pt_inter), (D,)
            for n in ("norm1", "norm2"):
                s[p + f"{n}.weight"], s[p + f"{n}.bias"] = (D,), (D,)
        s["metric_head.patch_transformer.embedding_convPxP.weight"] = (D, B, 1, 1)
        s["metric_head.patch_transformer.embedding_convPxP.bias"] = (D,)
        s["metric_head.mlp_classifier.linear1.weight"], s["metric_head.mlp_classifier.linear1.bias"] = (128, 128), (128,)
        s["metric_head.mlp_classifier.linear2.weight"], s["metric_head.mlp_classifier.linear2.bias"] = (2, 128), (2,)
    for name in cfg.head_names:
        p = "metric_head.seed_bin_regressor." if single else f"metric_head.seed_bin_regressors.{name}."
        s[p + "conv1.weight"], s[p + "conv1.bias"] = (SM, B, 1, 1), (SM,)
        s[p + "conv2.weight"], s[p + "conv2.bias"] = (cfg.n_bins, SM, 1, 1), (cfg.n_bins,)
    s["metric_head.seed_projector.conv1.weight"], s["metric_head.seed_projector.conv1.bias"] = (PM, B, 1, 1), (PM,)
    s["metric_head.seed_projector.conv2.weight"], s["metric_head.seed_projector.conv2.bias"] = (E, PM, 1, 1), (E,)
    for i in range(4):
        p = f"metric_head.projectors.{i}."
        s[p + "conv1.weight"], s[p + "conv1.bias"] = (PM, F_, 1, 1), (PM,)
        s[p + "conv2.weight"], s[p + "conv2.bias"] = (E, PM, 1, 1), (E,)
    cin = cfg.rel_features + (1 if single else 0) + E
    for name in cfg.head_names:
        mid = "" if single else f"{name}."
        for i in range(4):
            p = f"metric_head.attractors.{mid}{i}."
            s[p + "conv1.weight"], s[p + "conv1.bias"] = (E, E, 1, 1), (E,)
            s[p + "conv2.weight"], s[p + "conv2.bias"] = (cfg.attractors_at(i), E, 1, 1), (cfg.attractors_at(i),)
        p = f"metric_head.conditional_log_binomial.{mid}mlp."
        s[p + "0.weight"], s[p + "0.bias"] = (HID, cin, 1, 1), (HID,)
        s[p + "2.weight"], s[p + "2.bias"] = (4, HID, 1, 1), (4,)
    return s


def random_zoedepth_weights(cfg, seed: int = 0) -> dict:
    """Random-init fp32 CPU weights with activations kept O(1) through the net (fan-in scaled GEMM/conv
    weights, LayerNorm ~ identity, layer-scale ~0.1, a non-zero relative-position table, ordered seed bins)."""
    import math
    import torch
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in zoedepth_param_shapes(cfg).items():
        x = torch.randn(shape, generator=g)
        leaf = name.rsplit(".", 1)[-1]
        if "lambda_" in name:
            x = 0.1 * (1.0 + 0.2 * x)
        elif "relative_position_bias_table" in name or "cls_token" in name:
            x = 0.5 * x
        elif "layernorm" in name or ".norm1." in name or ".norm2." in name:
            x = (1.0 + 0.1 * x) if leaf == "weight" else 0.1 * x
        elif "seed_bin_regressors" in name and name.endswith("conv2.bias"):
            x = torch.linspace(-3.0, 3.0, x.numel()) + 0.1 * x
        elif leaf == "bias":
            x = 0.1 * x
        else:
            fan_in = int(np.prod(shape[1:]))
            if "reassemble_stage.layers" in name and "resize.weight" in name and shape[2] in (2, 4):
                fan_in = shape[0]
            gain = 4.0 if (("metric_head.attractors" in name and ".conv2." in name) or
                           ("conditional_log_binomial" in name and ".mlp.2." in name)) else 1.0
            x = x * (gain / math.sqrt(fan_in))
        out[name] = x.contiguous()
    return out


def random_cyclepose_weights(seed: int = 0) -> dict:
    import math
    import torch
    shapes = {
        "initial_model.1.weight": (64, 6, 7, 7), "initial_model.1.bias": (64,),
        "downsampling.0.weight": (128, 64, 3, 3), "downsampling.0.bias": (128,),
        "downsampling.3.weight": (256, 128, 3, 3), "downsampling.3.bias": (256,),
        "pose_conv.0.weight": (512, 256, 3, 3), "pose_conv.0.bias": (512,),
        "pose_dense.1.weight": (128, 512), "pose_dense.1.bias": (128,),
        "pose_dense.3.weight": (7, 128), "pose_dense.3.bias": (7,),
        "skip_linear.weight": (7, 512 + 256 * 32 * 32), "skip_linear.bias": (7,),
    }
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in shapes.items():
        x = torch.randn(shape, generator=g)
        out[name] = (0.1 * x if name.endswith("bias") else x / math.sqrt(int(np.prod(shape[1:])))).contiguous()
    out["pose_dense.3.bias"][3] += 4.0      # keep the quaternion near identity (small inter-frame rotations)
    return out


# ---------------------------------------------------------------------------------------------------
# Weight statistics that do not look like the seeded Gaussian ones (trained BEiT-L checkpoints carry a per-channel layer-scale
# spanning decades, a few outlier channels behind the LayerNorms and heavy-tailed weights).  Each edits a random_zoedepth_weights /
# oracle synth_weights dict IN PLACE; tests/test_zoedepth_gpu.py holds the 1e-4 m tolerance on each, bench.py --weights outlier
# measures the throughput of what the calibration has to switch on for the outlier set.
# ---------------------------------------------------------------------------------------------------
def layerscale_wide(w):
    import torch
    g = torch.Generator().manual_seed(101)
    for k in w:
        if k.endswith("lambda_1") or k.endswith("lambda_2"):          # log-uniform 1e-3 .. 1 per channel (trained BEiT: 1e-5 init, grown unevenly)
            w[k] = w[k].sign() * torch.pow(10.0, -3.0 * torch.rand(w[k].shape, generator=g)) * 0.3


def outlier_channels(w):
    import torch
    g = torch.Generator().manual_seed(102)
    idx = torch.randperm(1024, generator=g)[:6]
    for k in w:
        if k.endswith("layernorm_before.weight") or k.endswith("layernorm_after.weight"):       # 6 channels 50x larger after every LayerNorm
            w[k] = w[k].clone()
            w[k][idx] *= 50.0
        if k.endswith("attention.q_proj.weight") or k.endswith("mlp.fc1.weight"):               # ... and damped again where they are consumed, so
            w[k] = w[k].clone()                                                                     # the network stays in range
            w[k][:, idx] /= 25.0


def heavy_tailed(w):
    import torch
    g = torch.Generator().manual_seed(103)
    for k in w:
        if w[k].dim() >= 2 and w[k].numel() >= 1 << 16 and "position_bias" not in k:
            t = torch.randn(w[k].shape, generator=g) / torch.randn(w[k].shape, generator=g).abs().clamp_min(0.35)   # ratio of normals: heavy tails
            w[k] = w[k] * (0.6 + 0.4 * t.abs().clamp_max(12.0) / 1.6)


WEIGHT_VARIANTS = {"gaussian": None, "outlier": outlier_channels, "layerscale": layerscale_wide, "heavytail": heavy_tailed}
