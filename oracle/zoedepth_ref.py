"""CPU oracle for the MDEM hot path (ZoeDepth ZoeD_NK) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the product path (bodyslam_amd/) never does.

What it restates
----------------
The reference calls ``torch.hub.load("isl-org/ZoeDepth", "ZoeD_NK").infer_pil(image,
output_type="pil")`` (BodySLAM_Refactored/src/depth_estimation/interface.py:46,61;
BodySLAM_not_refactored/MDEM/mdem_interface.py:37-44,68).  That arithmetic lives in un-vendored
third parties (isl-org/ZoeDepth @ unpinned hub branch, intel-isl/MiDaS, timm==0.6.7); none is
under /root/reference.  This file is a plain-torch fp32 functional restatement of the published
algorithm, written against the weight-compatible restatement that IS installed in the image
(HF transformers 5.15.0, models/zoedepth/modeling_zoedepth.py + models/beit/modeling_beit.py +
models/zoedepth/image_processing_pil_zoedepth.py, which cites upstream commit edb6daf4).
Parameter names are the HF state_dict names so an ``Intel/zoedepth-nyu-kitti`` checkpoint loads
unchanged.

Pinning
-------
tests/test_oracle_zoedepth.py checks this file (a) live against HF ``ZoeDepthForDepthEstimation``
with identical seeded weights (tiny and full-size configs) and (b) against committed golden vectors
in tests/golden/ produced by oracle/make_golden.py from HF.  The reference's own golden pair
(tests/resources/depth_estimation/input_image.jpg -> output_depth_map.png) needs the real
pretrained weights, which are not available offline: for real-weights parity the oracle is
"parity unpinned"; for same-weights-in/same-numbers-out it is pinned against HF.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------
@dataclass
class ZoeConfig:
    """Subset of HF ZoeDepthConfig / BeitConfig that the forward depends on."""
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    intermediate: int = 4096
    taps: Tuple[int, ...] = (6, 12, 18, 24)
    image_size: int = 384            # pre-training image size -> rel-pos table is (2*24-1)^2+3
    patch: int = 16
    ln_eps: float = 1e-12
    neck_hidden: Tuple[int, ...] = (256, 512, 1024, 1024)
    fusion: int = 256
    reassemble_factors: Tuple[float, ...] = (4, 2, 1, 0.5)
    rel_features: int = 32
    bottleneck: int = 256
    bin_dim: int = 128
    n_attractors: int = 16           # NK head: every attractor layer emits 16 (modeling_zoedepth.py:1026-1031,:670)
    # single-head models (ZoeD_N, ZoeD_K = HF ZoeDepthMetricDepthEstimationHead, modeling_zoedepth.py:1106-1200): one bin
    # configuration, no router, per-level attractor counts, and the relative depth as a 33rd input of the log-binomial MLP
    level_attractors: Tuple[int, ...] = (16, 8, 4, 1)
    n_bins: int = 64
    min_temp: float = 0.0212
    max_temp: float = 50.0
    pt_layers: int = 4
    pt_hidden: int = 128
    pt_inter: int = 1024
    pt_heads: int = 4
    head_names: Tuple[str, ...] = ("nyu", "kitti")
    # HF ZoeDepthConfig.add_projection (modeling_zoedepth.py:344-346,358-360): a 3x3 256->256 conv + ReLU in front of the relative
    # head.  The survey's configuration ("as recalled") has it; HF's config default is False and upstream
    # MiDaS has no such layer, so a real checkpoint may lack it: the forward keys on the weight being present.
    add_projection: bool = True

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def single_head(self) -> bool:
        return len(self.head_names) == 1

    @property
    def seed_mlp(self) -> int:
        """hidden width of the seed bin regressor: HF's default 256 in the single head, bin_dim // 2 in the NK head (:1000)"""
        return 256 if self.single_head else self.bin_dim // 2

    @property
    def proj_mlp(self) -> int:
        """hidden width of the (seed) projectors: HF's default 128 in the single head, bin_dim // 2 in the NK head (:1003-1011)"""
        return 128 if self.single_head else self.bin_dim // 2

    def attractors_at(self, level: int) -> int:
        return self.level_attractors[level] if self.single_head else self.n_attractors


ZOED_NK = ZoeConfig()
ZOED_N = ZoeConfig(head_names=("nyu",))       # Intel/zoedepth-nyu   (max_depth 10; the softplus bin centres ignore it)
ZOED_K = ZoeConfig(head_names=("kitti",))     # Intel/zoedepth-kitti (max_depth 80)


def tiny_config() -> ZoeConfig:
    """A channel/depth-reduced backbone with the full-size neck/head: used for fast CPU tests."""
    return ZoeConfig(hidden=64, layers=4, heads=2, intermediate=128, taps=(1, 2, 3, 4), image_size=64)


# --------------------------------------------------------------------------------------------
# deterministic synthetic weights (no checkpoint is available offline)
# --------------------------------------------------------------------------------------------
def param_shapes(cfg: ZoeConfig) -> Dict[str, Tuple[int, ...]]:
    """HF state_dict names -> shapes for a ZoeD_NK-style model of configuration ``cfg``."""
    s: Dict[str, Tuple[int, ...]] = {}
    H, I = cfg.hidden, cfg.intermediate
    s["backbone.beit.embeddings.cls_token"] = (1, 1, H)
    s["backbone.beit.embeddings.patch_embeddings.projection.weight"] = (H, 3, cfg.patch, cfg.patch)
    s["backbone.beit.embeddings.patch_embeddings.projection.bias"] = (H,)
    win = cfg.image_size // cfg.patch
    nrd = (2 * win - 1) ** 2 + 3
    for l in range(cfg.layers):
        p = f"backbone.beit.layers.{l}."
        s[p + "lambda_1"] = (H,)
        s[p + "lambda_2"] = (H,)
        s[p + "attention.q_proj.weight"] = (H, H)
        s[p + "attention.q_proj.bias"] = (H,)
        s[p + "attention.k_proj.weight"] = (H, H)
        s[p + "attention.v_proj.weight"] = (H, H)
        s[p + "attention.v_proj.bias"] = (H,)
        s[p + "attention.o_proj.weight"] = (H, H)
        s[p + "attention.o_proj.bias"] = (H,)
        s[p + "layernorm_before.weight"] = (H,)
        s[p + "layernorm_before.bias"] = (H,)
        s[p + "layernorm_after.weight"] = (H,)
        s[p + "layernorm_after.bias"] = (H,)
        s[p + "mlp.fc1.weight"] = (I, H)
        s[p + "mlp.fc1.bias"] = (I,)
        s[p + "mlp.fc2.weight"] = (H, I)
        s[p + "mlp.fc2.bias"] = (H,)
        s[p + "relative_position_bias.relative_position_bias_table"] = (nrd, cfg.heads)
    for i, (c, f) in enumerate(zip(cfg.neck_hidden, cfg.reassemble_factors)):
        p = f"neck.reassemble_stage.layers.{i}."
        s[p + "projection.weight"] = (c, H, 1, 1)
        s[p + "projection.bias"] = (c,)
        if f > 1:
            s[p + "resize.weight"] = (c, c, int(f), int(f))
            s[p + "resize.bias"] = (c,)
        elif f < 1:
            s[p + "resize.weight"] = (c, c, 3, 3)
            s[p + "resize.bias"] = (c,)
        s[f"neck.reassemble_stage.readout_projects.{i}.0.weight"] = (H, 2 * H)
        s[f"neck.reassemble_stage.readout_projects.{i}.0.bias"] = (H,)
        s[f"neck.convs.{i}.weight"] = (cfg.fusion, c, 3, 3)
    Fh = cfg.fusion
    for i in range(4):
        p = f"neck.fusion_stage.layers.{i}."
        s[p + "projection.weight"] = (Fh, Fh, 1, 1)
        s[p + "projection.bias"] = (Fh,)
        for r in ("residual_layer1", "residual_layer2"):
            for c in ("convolution1", "convolution2"):
                s[p + f"{r}.{c}.weight"] = (Fh, Fh, 3, 3)
                s[p + f"{r}.{c}.bias"] = (Fh,)
    if cfg.add_projection:
        s["relative_head.projection.weight"] = (256, 256, 3, 3)
        s["relative_head.projection.bias"] = (256,)
    s["relative_head.conv1.weight"] = (Fh // 2, Fh, 3, 3)
    s["relative_head.conv1.bias"] = (Fh // 2,)
    s["relative_head.conv2.weight"] = (cfg.rel_features, Fh // 2, 3, 3)
    s["relative_head.conv2.bias"] = (cfg.rel_features,)
    s["relative_head.conv3.weight"] = (1, cfg.rel_features, 1, 1)
    s["relative_head.conv3.bias"] = (1,)
    B, E = cfg.bottleneck, cfg.bin_dim
    s["metric_head.conv2.weight"] = (B, B, 1, 1)
    s["metric_head.conv2.bias"] = (B,)
    if cfg.single_head:
        p = "metric_head.seed_bin_regressor."
        s[p + "conv1.weight"] = (cfg.seed_mlp, B, 1, 1)
        s[p + "conv1.bias"] = (cfg.seed_mlp,)
        s[p + "conv2.weight"] = (cfg.n_bins, cfg.seed_mlp, 1, 1)
        s[p + "conv2.bias"] = (cfg.n_bins,)
    else:
        for l in range(cfg.pt_layers):
            p = f"metric_head.patch_transformer.transformer_encoder.{l}."
            for n in ("query", "key", "value", "out_proj"):
                s[p + f"self_attn.{n}.weight"] = (cfg.pt_hidden, cfg.pt_hidden)
                s[p + f"self_attn.{n}.bias"] = (cfg.pt_hidden,)
            s[p + "linear1.weight"] = (cfg.pt_inter, cfg.pt_hidden)
            s[p + "linear1.bias"] = (cfg.pt_inter,)
            s[p + "linear2.weight"] = (cfg.pt_hidden, cfg.pt_inter)
            s[p + "linear2.bias"] = (cfg.pt_hidden,)
            for n in ("norm1", "norm2"):
                s[p + f"{n}.weight"] = (cfg.pt_hidden,)
                s[p + f"{n}.bias"] = (cfg.pt_hidden,)
        s["metric_head.patch_transformer.embedding_convPxP.weight"] = (cfg.pt_hidden, B, 1, 1)
        s["metric_head.patch_transformer.embedding_convPxP.bias"] = (cfg.pt_hidden,)
        s["metric_head.mlp_classifier.linear1.weight"] = (128, 128)
        s["metric_head.mlp_classifier.linear1.bias"] = (128,)
        s["metric_head.mlp_classifier.linear2.weight"] = (2, 128)
        s["metric_head.mlp_classifier.linear2.bias"] = (2,)
        for name in cfg.head_names:
            p = f"metric_head.seed_bin_regressors.{name}."
            s[p + "conv1.weight"] = (E // 2, B, 1, 1)
            s[p + "conv1.bias"] = (E // 2,)
            s[p + "conv2.weight"] = (cfg.n_bins, E // 2, 1, 1)
            s[p + "conv2.bias"] = (cfg.n_bins,)
    PM = cfg.proj_mlp
    s["metric_head.seed_projector.conv1.weight"] = (PM, B, 1, 1)
    s["metric_head.seed_projector.conv1.bias"] = (PM,)
    s["metric_head.seed_projector.conv2.weight"] = (E, PM, 1, 1)
    s["metric_head.seed_projector.conv2.bias"] = (E,)
    for i in range(4):
        p = f"metric_head.projectors.{i}."
        s[p + "conv1.weight"] = (PM, Fh, 1, 1)
        s[p + "conv1.bias"] = (PM,)
        s[p + "conv2.weight"] = (E, PM, 1, 1)
        s[p + "conv2.bias"] = (E,)
    for name in cfg.head_names:
        mid = "" if cfg.single_head else f"{name}."
        for i in range(4):
            p = f"metric_head.attractors.{mid}{i}."
            s[p + "conv1.weight"] = (E, E, 1, 1)
            s[p + "conv1.bias"] = (E,)
            s[p + "conv2.weight"] = (cfg.attractors_at(i), E, 1, 1)
            s[p + "conv2.bias"] = (cfg.attractors_at(i),)
        p = f"metric_head.conditional_log_binomial.{mid}mlp."
        cin = cfg.rel_features + (1 if cfg.single_head else 0) + E      # single head: + the relative depth (:1156)
        hid = cin // (2 if cfg.single_head else 4)                       # bottleneck_factor: default 2 (:1159) vs 4 in the NK head
        s[p + "0.weight"] = (hid, cin, 1, 1)
        s[p + "0.bias"] = (hid,)
        s[p + "2.weight"] = (4, hid, 1, 1)
        s[p + "2.bias"] = (4,)
    return s


def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def synth_weights(cfg: ZoeConfig, seed: int = 0, route_bias: float = 0.0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic fp32 weights (per-tensor generator keyed by name + seed).

    Scales are chosen so every block is exercised with O(1) activations: fan-in-normalised
    GEMM/conv weights, non-trivial LayerNorm affine, layer-scale ~0.1 (the BEiT-L init value),
    a non-zero relative-position-bias table.  ``route_bias`` is added to the domain classifier's
    logit 0 minus logit 1 to force a route in tests.
    """
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        rng = np.random.default_rng(_name_seed(name, seed))
        n = int(np.prod(shape))
        x = rng.standard_normal(n, dtype=np.float32).reshape(shape)
        leaf = name.rsplit(".", 1)[-1]
        if "lambda_" in name:
            x = 0.1 * (1.0 + 0.2 * x)
        elif "relative_position_bias_table" in name:
            x = 0.5 * x
        elif "cls_token" in name:
            x = 0.5 * x
        elif "layernorm" in name or ".norm1." in name or ".norm2." in name:
            x = (1.0 + 0.1 * x) if leaf == "weight" else 0.1 * x
        elif "seed_bin_regressors" in name and name.endswith("conv2.bias"):
            # ordered seed bins (softplus of an increasing ramp) so that the log-binomial mode
            # moves the depth monotonically, as with trained weights
            x = np.linspace(-3.0, 3.0, n, dtype=np.float32) + 0.1 * x
        elif leaf == "bias":
            x = 0.1 * x
        else:  # GEMM / conv weight: std = gain / sqrt(fan_in)
            fan_in = int(np.prod(shape[1:]))
            if "reassemble_stage.layers" in name and "resize.weight" in name and len(shape) == 4 and shape[2] in (2, 4):
                fan_in = shape[0]  # ConvTranspose2d weight is [C_in, C_out, k, k]; each output sees C_in taps
            gain = 1.0
            if "metric_head.attractors" in name and ".conv2." in name:
                gain = 4.0   # spread the attractor points over a few metres
            if "conditional_log_binomial" in name and ".mlp.2." in name:
                gain = 4.0   # make p and the temperature vary per pixel
            x = x * (gain / math.sqrt(fan_in))
        out[name] = torch.from_numpy(np.ascontiguousarray(x))
    if route_bias != 0.0:
        b = out["metric_head.mlp_classifier.linear2.bias"].clone()
        b[0] += route_bias
        b[1] -= route_bias
        out["metric_head.mlp_classifier.linear2.bias"] = b
    return out


# --------------------------------------------------------------------------------------------
# pre / post processing (HF image_processing_pil_zoedepth.py:72-108,133-232,234-341; upstream
# depth_model.py#L57 infer_pil: pad -> resize -> normalise -> forward(+flip) -> resize -> crop)
# --------------------------------------------------------------------------------------------
def pad_sizes(h: int, w: int) -> Tuple[int, int]:
    return int(np.sqrt(h / 2) * 3), int(np.sqrt(w / 2) * 3)


def net_size(h_padded: int, w_padded: int, out_hw=(384, 512), multiple: int = 32) -> Tuple[int, int]:
    """keep_aspect_ratio=True, ensure_multiple_of=32 (image_processing_pil_zoedepth.py:72-108)."""
    sh, sw = out_hw[0] / h_padded, out_hw[1] / w_padded
    if abs(1 - sw) < abs(1 - sh):
        sh = sw
    else:
        sw = sh

    def con(v):
        return int(np.round(v / multiple) * multiple)

    return con(sh * h_padded), con(sw * w_padded)


def preprocess(frames_u8: torch.Tensor, out_hw=(384, 512)) -> torch.Tensor:
    """uint8 [B,H,W,3] -> float32 [B,3,h,w] network input (rescale, reflect pad, bilinear
    align_corners=True resize, normalise mean=std=0.5).  out_hw is the processor's target size
    (384x512 for every released checkpoint; tests use smaller targets to run quickly)."""
    x = frames_u8.permute(0, 3, 1, 2).to(torch.float32) * (1.0 / 255.0)
    H, W = x.shape[-2:]
    ph, pw = pad_sizes(H, W)
    x = F.pad(x, (pw, pw, ph, ph), mode="reflect")
    nh, nw = net_size(H + 2 * ph, W + 2 * pw, out_hw)
    x = F.interpolate(x, (nh, nw), mode="bilinear", align_corners=True)
    return (x - 0.5) / 0.5


def postprocess(depth: torch.Tensor, depth_flipped: Optional[torch.Tensor], H: int, W: int) -> torch.Tensor:
    """[B,h,w] (+ flipped forward) -> [B,H,W] float32 metres: un-flip + average, bicubic
    (align_corners=False, no antialias) to the padded size, crop the padding."""
    if depth_flipped is not None:
        depth = (depth + torch.flip(depth_flipped, dims=[-1])) / 2
    ph, pw = pad_sizes(H, W)
    d = F.interpolate(depth.unsqueeze(1), (H + 2 * ph, W + 2 * pw), mode="bicubic", align_corners=False)
    d = d[:, 0]
    if ph > 0:
        d = d[:, ph:-ph, :]
    if pw > 0:
        d = d[:, :, pw:-pw]
    return d.contiguous()


def to_uint16(depth_m: torch.Tensor) -> np.ndarray:
    """upstream infer_pil(output_type='pil'): (depth*256).astype(uint16) -> PIL 'I;16'."""
    return (depth_m.numpy() * 256.0).astype(np.uint16)


# --------------------------------------------------------------------------------------------
# BEiT backbone (modeling_beit.py:63-176 embeddings, :179-265 rel-pos bias, :296-341 attention,
# :344-357 MLP, :384-444 layer)
# --------------------------------------------------------------------------------------------
def relative_position_index(wh: int, ww: int) -> torch.Tensor:
    nrd = (2 * wh - 1) * (2 * ww - 1) + 3
    coords = torch.stack(torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wh - 1
    rel[:, :, 1] += ww - 1
    rel[:, :, 0] *= 2 * ww - 1
    idx = torch.zeros((wh * ww + 1,) * 2, dtype=rel.dtype)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = nrd - 3
    idx[0:, 0] = nrd - 2
    idx[0, 0] = nrd - 1
    return idx


def relative_position_bias(table: torch.Tensor, old_win: int, wh: int, ww: int) -> torch.Tensor:
    """table [(2*old-1)^2+3, heads] -> bias [heads, T, T] for a (wh, ww) token window
    (bilinear re-interpolation of the table, modeling_beit.py:220-265)."""
    old = 2 * old_win - 1
    nh, nw = 2 * wh - 1, 2 * ww - 1
    sub = table[: old * old].reshape(1, old, old, -1).permute(0, 3, 1, 2)
    new = F.interpolate(sub, size=(nh, nw), mode="bilinear")
    new = new.permute(0, 2, 3, 1).reshape(nh * nw, -1)
    full = torch.cat([new, table[old * old:]])
    idx = relative_position_index(wh, ww)
    T = wh * ww + 1
    return full[idx.view(-1)].view(T, T, -1).permute(2, 0, 1).contiguous()


def beit_forward(w: Dict[str, torch.Tensor], cfg: ZoeConfig, x: torch.Tensor,
                 taps_out: Optional[dict] = None) -> List[torch.Tensor]:
    """x [B,3,h,w] -> hidden states after the tap layers, each [B, 1+hp*wp, hidden]."""
    B, _, h, wd = x.shape
    hp, wp = h // cfg.patch, wd // cfg.patch
    pe = "backbone.beit.embeddings."
    t = F.conv2d(x, w[pe + "patch_embeddings.projection.weight"], w[pe + "patch_embeddings.projection.bias"],
                 stride=cfg.patch).flatten(2).transpose(1, 2)
    t = torch.cat([w[pe + "cls_token"].expand(B, -1, -1), t], dim=1)
    if taps_out is not None:
        taps_out["embed"] = t
    nh, hd = cfg.heads, cfg.head_dim
    outs = []
    for l in range(cfg.layers):
        p = f"backbone.beit.layers.{l}."
        bias = relative_position_bias(w[p + "relative_position_bias.relative_position_bias_table"],
                                      cfg.image_size // cfg.patch, hp, wp)
        y = F.layer_norm(t, (cfg.hidden,), w[p + "layernorm_before.weight"], w[p + "layernorm_before.bias"], cfg.ln_eps)
        q = F.linear(y, w[p + "attention.q_proj.weight"], w[p + "attention.q_proj.bias"])
        k = F.linear(y, w[p + "attention.k_proj.weight"])
        v = F.linear(y, w[p + "attention.v_proj.weight"], w[p + "attention.v_proj.bias"])
        q = q.view(B, -1, nh, hd).transpose(1, 2)
        k = k.view(B, -1, nh, hd).transpose(1, 2)
        v = v.view(B, -1, nh, hd).transpose(1, 2)
        a = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5) + bias.unsqueeze(0)
        a = torch.softmax(a, dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, -1, cfg.hidden)
        o = F.linear(o, w[p + "attention.o_proj.weight"], w[p + "attention.o_proj.bias"])
        t = w[p + "lambda_1"] * o + t
        y = F.layer_norm(t, (cfg.hidden,), w[p + "layernorm_after.weight"], w[p + "layernorm_after.bias"], cfg.ln_eps)
        y = F.gelu(F.linear(y, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]))
        y = F.linear(y, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        t = w[p + "lambda_2"] * y + t
        if taps_out is not None:
            taps_out[f"layer{l + 1}"] = t
        if (l + 1) in cfg.taps:
            outs.append(t)
    return outs


# --------------------------------------------------------------------------------------------
# DPT neck (modeling_zoedepth.py:55-149 reassemble, :153-329 fusion) + relative head (:332-373)
# --------------------------------------------------------------------------------------------
def _res_unit(w, p, x):
    y = F.conv2d(F.relu(x), w[p + "convolution1.weight"], w[p + "convolution1.bias"], padding=1)
    y = F.conv2d(F.relu(y), w[p + "convolution2.weight"], w[p + "convolution2.bias"], padding=1)
    return y + x


def neck_forward(w, cfg: ZoeConfig, hiddens: Sequence[torch.Tensor], hp: int, wp: int, taps_out=None):
    B = hiddens[0].shape[0]
    feats = []
    for i, (hs, f) in enumerate(zip(hiddens, cfg.reassemble_factors)):
        cls, tok = hs[:, 0], hs[:, 1:]
        y = torch.cat([tok, cls.unsqueeze(1).expand_as(tok)], dim=-1)
        y = F.gelu(F.linear(y, w[f"neck.reassemble_stage.readout_projects.{i}.0.weight"],
                            w[f"neck.reassemble_stage.readout_projects.{i}.0.bias"]))
        y = y.permute(0, 2, 1).reshape(B, -1, hp, wp)
        p = f"neck.reassemble_stage.layers.{i}."
        y = F.conv2d(y, w[p + "projection.weight"], w[p + "projection.bias"])
        if f > 1:
            y = F.conv_transpose2d(y, w[p + "resize.weight"], w[p + "resize.bias"], stride=int(f))
        elif f < 1:
            y = F.conv2d(y, w[p + "resize.weight"], w[p + "resize.bias"], stride=int(1 / f), padding=1)
        if taps_out is not None:
            taps_out[f"reassemble{i}"] = y
        y = F.conv2d(y, w[f"neck.convs.{i}.weight"], None, padding=1)
        if taps_out is not None:
            taps_out[f"neckconv{i}"] = y
        feats.append(y)
    fused_list = []
    fused = None
    for li, feat in enumerate(feats[::-1]):
        p = f"neck.fusion_stage.layers.{li}."
        if fused is None:
            fused = feat
        else:
            if fused.shape != feat.shape:
                feat = F.interpolate(feat, size=fused.shape[2:], mode="bilinear", align_corners=False)
            fused = fused + _res_unit(w, p + "residual_layer1.", feat)
        fused = _res_unit(w, p + "residual_layer2.", fused)
        fused = F.interpolate(fused, scale_factor=2, mode="bilinear", align_corners=True)
        fused = F.conv2d(fused, w[p + "projection.weight"], w[p + "projection.bias"])
        if taps_out is not None:
            taps_out[f"fused{li}"] = fused
        fused_list.append(fused)
    return fused_list, feats[-1]


def relative_head_forward(w, fused_last: torch.Tensor, taps_out=None):
    y = fused_last
    if "relative_head.projection.weight" in w:       # config.add_projection (modeling_zoedepth.py:358-360)
        y = F.relu(F.conv2d(y, w["relative_head.projection.weight"], w["relative_head.projection.bias"], padding=1))
    y = F.conv2d(y, w["relative_head.conv1.weight"], w["relative_head.conv1.bias"], padding=1)
    y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
    feat = F.relu(F.conv2d(y, w["relative_head.conv2.weight"], w["relative_head.conv2.bias"], padding=1))
    rel = F.relu(F.conv2d(feat, w["relative_head.conv3.weight"], w["relative_head.conv3.bias"]))
    if taps_out is not None:
        taps_out["rel_features"] = feat
    return rel[:, 0], feat


# --------------------------------------------------------------------------------------------
# metric bins head, NK variant (modeling_zoedepth.py:965-1103; attractor :665-746; seed :494-547;
# projector :749-772; log-binomial :376-491; router :885-962)
# --------------------------------------------------------------------------------------------
def _c1(w, name, x):
    return F.conv2d(x, w[name + ".weight"], w[name + ".bias"])


def _inv_attractor(dx: torch.Tensor, alpha: float = 300.0, gamma: int = 2) -> torch.Tensor:
    # called with its defaults at modeling_zoedepth.py:741 (config.attractor_alpha is NOT forwarded)
    return dx.div(1 + alpha * dx.pow(gamma))


def router_logits(w, cfg: ZoeConfig, x: torch.Tensor) -> torch.Tensor:
    """x = metric_head.conv2(bottleneck) [B,256,h,w] -> domain logits [B,2]."""
    e = _c1(w, "metric_head.patch_transformer.embedding_convPxP", x).flatten(2)
    e = F.pad(e, (1, 0)).permute(0, 2, 1)
    B, S, D = e.shape
    pos = torch.arange(0, S, dtype=e.dtype).unsqueeze(1)
    idx = torch.arange(0, D, 2, dtype=e.dtype).unsqueeze(0)
    div = torch.exp(idx * (-torch.log(torch.full((), 10000.0)) / D))
    pe = pos * div
    e = e + torch.cat([torch.sin(pe), torch.cos(pe)], dim=1).unsqueeze(0)
    nh = cfg.pt_heads
    hd = D // nh
    for l in range(cfg.pt_layers):
        p = f"metric_head.patch_transformer.transformer_encoder.{l}."
        q = F.linear(e, w[p + "self_attn.query.weight"], w[p + "self_attn.query.bias"]).view(B, S, nh, hd).transpose(1, 2)
        k = F.linear(e, w[p + "self_attn.key.weight"], w[p + "self_attn.key.bias"]).view(B, S, nh, hd).transpose(1, 2)
        v = F.linear(e, w[p + "self_attn.value.weight"], w[p + "self_attn.value.bias"]).view(B, S, nh, hd).transpose(1, 2)
        a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        o = torch.matmul(a, v).permute(0, 2, 1, 3).reshape(B, S, D)
        o = F.linear(o, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        e = F.layer_norm(e + o, (D,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
        o = F.linear(F.relu(F.linear(e, w[p + "linear1.weight"], w[p + "linear1.bias"])),
                     w[p + "linear2.weight"], w[p + "linear2.bias"])
        e = F.layer_norm(e + o, (D,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)
    emb = e[:, 0, :]
    y = F.relu(F.linear(emb, w["metric_head.mlp_classifier.linear1.weight"], w["metric_head.mlp_classifier.linear1.bias"]))
    return F.linear(y, w["metric_head.mlp_classifier.linear2.weight"], w["metric_head.mlp_classifier.linear2.bias"])


def metric_head_single(w, cfg: ZoeConfig, name: str, x: torch.Tensor, blocks: Sequence[torch.Tensor],
                       last: torch.Tensor, taps_out=None, rel: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One named head on a batch that was routed to it.  x = conv2(bottleneck).  Single-head models (ZoeD_N / ZoeD_K,
    modeling_zoedepth.py:1166-1200) use un-named parameters and feed the relative depth ``rel`` [B,h,w] to the last MLP."""
    mid = "" if cfg.single_head else f"{name}."
    p = "metric_head.seed_bin_regressor." if cfg.single_head else f"metric_head.seed_bin_regressors.{name}."
    seed = F.softplus(_c1(w, p + "conv2", F.relu(_c1(w, p + "conv1", x))))
    prev_bin = seed
    prev_emb = _c1(w, "metric_head.seed_projector.conv2", F.relu(_c1(w, "metric_head.seed_projector.conv1", x)))
    bin_centers = None
    emb = None
    for i, feat in enumerate(blocks):
        pp = f"metric_head.projectors.{i}."
        emb = _c1(w, pp + "conv2", F.relu(_c1(w, pp + "conv1", feat)))
        pa = f"metric_head.attractors.{mid}{i}."
        y = emb + F.interpolate(prev_emb, emb.shape[-2:], mode="bilinear", align_corners=True)
        A = F.softplus(_c1(w, pa + "conv2", F.relu(_c1(w, pa + "conv1", y))))
        c = F.interpolate(prev_bin, A.shape[-2:], mode="bilinear", align_corners=True)
        delta = torch.zeros_like(c)
        for a in range(cfg.attractors_at(i)):
            delta += _inv_attractor(A[:, a, ...].unsqueeze(1) - c)
        delta = delta / cfg.attractors_at(i)
        bin_centers = c + delta
        prev_bin = bin_centers
        prev_emb = emb
        if taps_out is not None:
            taps_out[f"bins{i}"] = bin_centers
    bc = F.interpolate(bin_centers, last.shape[-2:], mode="bilinear", align_corners=True)
    em = F.interpolate(emb, last.shape[-2:], mode="bilinear", align_corners=True)
    pm = f"metric_head.conditional_log_binomial.{mid}mlp."
    if cfg.single_head:      # :1186-1191: the relative depth (already at `last`'s size here) is concatenated to the features
        last = torch.cat([last, F.interpolate(rel.unsqueeze(1), size=last.shape[2:], mode="bilinear", align_corners=True)], dim=1)
    pt = F.softplus(_c1(w, pm + "2", F.gelu(_c1(w, pm + "0", torch.cat([last, em], dim=1)))))
    prob = pt[:, :2] + 1e-4
    prob = prob[:, 0] / (prob[:, 0] + prob[:, 1])
    temp = pt[:, 2:] + 1e-4
    temp = (temp[:, 0] / (temp[:, 0] + temp[:, 1])).unsqueeze(1)
    temp = (cfg.max_temp - cfg.min_temp) * temp + cfg.min_temp
    prob = prob.unsqueeze(1)
    eps = 1e-4
    omp = (1 - prob).clamp(min=eps, max=1.0)
    prob = prob.clamp(min=eps, max=1.0)
    kidx = torch.arange(0, cfg.n_bins, dtype=prob.dtype).view(1, -1, 1, 1)
    km1 = torch.tensor([cfg.n_bins - 1], dtype=prob.dtype).view(1, -1, 1, 1)

    def log_binom(n, k, e=1e-7):
        n = n + e
        k = k + e
        return n * torch.log(n) - k * torch.log(k) - (n - k) * torch.log(n - k + e)

    yk = log_binom(km1, kidx) + kidx * torch.log(prob) + (km1 - kidx) * torch.log(omp)
    px = torch.softmax(yk / temp, dim=1)
    return torch.sum(px * bc, dim=1)


def zoedepth_forward(w: Dict[str, torch.Tensor], cfg: ZoeConfig, x: torch.Tensor, taps_out: Optional[dict] = None,
                     per_image_route: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [B,3,h,w] normalised network input -> (metric depth [B,h,w], domain logits [B,2]).

    ``per_image_route=True`` routes every image by its own logits: that is what the reference
    computes, because it always calls the network with batch 1 (interface.py:61 -> infer_pil; the
    flip-aug pass is a second batch-1 call).  ``False`` reproduces HF's batch-summed vote
    (modeling_zoedepth.py:1063-1067) and is used only to pin this file against HF at B>1.
    """
    B, _, h, wd = x.shape
    hp, wp = h // cfg.patch, wd // cfg.patch
    hiddens = beit_forward(w, cfg, x, taps_out)
    fused, bottleneck = neck_forward(w, cfg, hiddens, hp, wp, taps_out)
    rel, last = relative_head_forward(w, fused[-1], taps_out)
    xb = _c1(w, "metric_head.conv2", bottleneck)
    if cfg.single_head:
        t = {} if taps_out is not None else None
        out = metric_head_single(w, cfg, cfg.head_names[0], xb, fused, last, t, rel=rel)
        if taps_out is not None:
            for k_, v_ in t.items():
                taps_out[f"{cfg.head_names[0]}.{k_}"] = (torch.arange(B), v_)
            taps_out["rel_depth"] = rel
            taps_out["depth_net"] = out
        return out, torch.zeros(B, 2, dtype=x.dtype)
    logits = router_logits(w, cfg, xb)
    if per_image_route:
        route = torch.argmax(logits, dim=-1)
    else:
        route = torch.argmax(torch.softmax(logits.sum(dim=0, keepdim=True), dim=-1), dim=-1).expand(B)
    out = torch.empty(B, h, wd, dtype=x.dtype)
    for r, name in enumerate(cfg.head_names):
        sel = (route == r).nonzero().flatten()
        if sel.numel() == 0:
            continue
        t = {} if taps_out is not None else None
        out[sel] = metric_head_single(w, cfg, name, xb[sel], [f[sel] for f in fused], last[sel], t)
        if taps_out is not None:
            for k_, v_ in t.items():
                taps_out[f"{name}.{k_}"] = (sel, v_)
    if taps_out is not None:
        taps_out["route"] = route
        taps_out["rel_depth"] = rel
        taps_out["depth_net"] = out
    return out, logits


def infer_depth(w: Dict[str, torch.Tensor], cfg: ZoeConfig, frames_u8: torch.Tensor, flip_aug: bool = True,
                chunk: int = 2, out_hw=(384, 512)) -> torch.Tensor:
    """The whole MDEM path on uint8 frames [B,H,W,3] -> float32 metres [B,H,W] (before the
    x256 -> uint16 quantisation of infer_pil)."""
    B, H, W, _ = frames_u8.shape
    outs = []
    with torch.no_grad():
        for s in range(0, B, chunk):
            x = preprocess(frames_u8[s:s + chunk], out_hw)
            d, _ = zoedepth_forward(w, cfg, x)
            df = None
            if flip_aug:
                df, _ = zoedepth_forward(w, cfg, torch.flip(x, dims=[3]))
            outs.append(postprocess(d, df, H, W))
    return torch.cat(outs, dim=0)
