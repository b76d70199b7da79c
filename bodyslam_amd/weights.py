"""Weight containers the hot path reads.

  * ZoeDepth.  The engine takes a state dict with the parameter names of the installed HF ``ZoeDepthForDepthEstimation``
    (transformers 5.x: ``backbone.beit.layers.N.attention.q_proj.weight`` ...).  ``load_zoedepth_weights`` accepts three
    containers and normalises their names:
      - a state dict with those names (.safetensors / .pt);
      - the published ``Intel/zoedepth-nyu-kitti`` (/-nyu, /-kitti) checkpoints, which carry the transformers 4.x names
        (``backbone.encoder.layer.N.attention.attention.query.weight`` ...; transformers 5 renames them on load through
        conversion_mapping.py "ViTModel" / "BeitModel" / "BeitBackbone" -- the same rules are applied here);
      - upstream ``ZoeD_M12_NK.pt`` (/_N, /_K) as ``torch.hub.load("isl-org/ZoeDepth", ...)`` downloads it -- what the reference's
        ``DepthEstimator`` actually loads (BodySLAM_Refactored/src/depth_estimation/interface.py:43-46): a dict with the model
        state under "model", names of isl-org/ZoeDepth + intel-isl/MiDaS + timm BEiT (``core.core.pretrained.model.blocks.N.attn.qkv
        .weight`` ...), fused q/k/v.  The rules below restate HF's convert_zoedepth_to_hf.py; no upstream checkpoint is reachable
        offline, so they are exercised by a round trip only (tests/test_weights_cpu.py) -- the loader fails loudly (KeyError with
        the offending names) on anything it does not know.
  * CyclePose: the reference's checkpoint container written by ModelIO.save_pose_model
    (BodySLAM_not_refactored/UTILS/io_utils.py:207-232): a dict with ``model_state_dict`` (+ epoch, ate, ...).
    ``skip_linear.*`` is read from it when present (quirk Q1, cyclepose.py).
No network access is attempted: the reference's torch.hub download is replaced by a local path, given explicitly or through
BODYSLAM_ZOEDEPTH_WEIGHTS / BODYSLAM_CYCLEPOSE_WEIGHTS.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Tuple

import torch

ENV_ZOE = "BODYSLAM_ZOEDEPTH_WEIGHTS"
ENV_POSE = "BODYSLAM_CYCLEPOSE_WEIGHTS"

# ------------------------------------------------------------------------------------------------------------------
# transformers 4.x names (the published hub checkpoints) -> the names used here (transformers 5.x modules)
# ------------------------------------------------------------------------------------------------------------------
_HF4_RULES: List[Tuple[str, str]] = [
    (r"^backbone\.encoder\.layer\.", "backbone.beit.layers."),
    (r"^backbone\.embeddings\.", "backbone.beit.embeddings."),
    (r"\.attention\.attention\.relative_position_bias\.", ".relative_position_bias."),
    (r"\.attention\.attention\.query\.", ".attention.q_proj."),
    (r"\.attention\.attention\.key\.", ".attention.k_proj."),
    (r"\.attention\.attention\.value\.", ".attention.v_proj."),
    (r"\.attention\.output\.dense\.", ".attention.o_proj."),
    (r"\.intermediate\.dense\.", ".mlp.fc1."),
    (r"\.output\.dense\.", ".mlp.fc2."),
]


def hf4_to_hf5(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if k.endswith("relative_position_index"):      # a buffer the forward recomputes (modeling_beit.py:194-218)
            continue
        if k.startswith("backbone."):
            for pat, rep in _HF4_RULES:
                k = re.sub(pat, rep, k)
        out[k] = v
    return out


# ------------------------------------------------------------------------------------------------------------------
# upstream isl-org/ZoeDepth (+ MiDaS DPT_BEiT_L_384, timm Beit) names <-> the names used here.
# Each rule is (upstream regex, HF template, HF regex, upstream template): applied to one key they are inverse to each other.
# ------------------------------------------------------------------------------------------------------------------
_B = r"core\.core\.pretrained\.model\."
_RULES: List[Tuple[str, str, str, str]] = [
    (rf"^{_B}cls_token$", "backbone.beit.embeddings.cls_token",
     r"^backbone\.beit\.embeddings\.cls_token$", "core.core.pretrained.model.cls_token"),
    (rf"^{_B}patch_embed\.proj\.(weight|bias)$", r"backbone.beit.embeddings.patch_embeddings.projection.\1",
     r"^backbone\.beit\.embeddings\.patch_embeddings\.projection\.(weight|bias)$", r"core.core.pretrained.model.patch_embed.proj.\1"),
    (rf"^{_B}blocks\.(\d+)\.gamma_([12])$", r"backbone.beit.layers.\1.lambda_\2",
     r"^backbone\.beit\.layers\.(\d+)\.lambda_([12])$", r"core.core.pretrained.model.blocks.\1.gamma_\2"),
    (rf"^{_B}blocks\.(\d+)\.norm1\.(weight|bias)$", r"backbone.beit.layers.\1.layernorm_before.\2",
     r"^backbone\.beit\.layers\.(\d+)\.layernorm_before\.(weight|bias)$", r"core.core.pretrained.model.blocks.\1.norm1.\2"),
    (rf"^{_B}blocks\.(\d+)\.norm2\.(weight|bias)$", r"backbone.beit.layers.\1.layernorm_after.\2",
     r"^backbone\.beit\.layers\.(\d+)\.layernorm_after\.(weight|bias)$", r"core.core.pretrained.model.blocks.\1.norm2.\2"),
    (rf"^{_B}blocks\.(\d+)\.attn\.proj\.(weight|bias)$", r"backbone.beit.layers.\1.attention.o_proj.\2",
     r"^backbone\.beit\.layers\.(\d+)\.attention\.o_proj\.(weight|bias)$", r"core.core.pretrained.model.blocks.\1.attn.proj.\2"),
    (rf"^{_B}blocks\.(\d+)\.attn\.relative_position_bias_table$", r"backbone.beit.layers.\1.relative_position_bias.relative_position_bias_table",
     r"^backbone\.beit\.layers\.(\d+)\.relative_position_bias\.relative_position_bias_table$",
     r"core.core.pretrained.model.blocks.\1.attn.relative_position_bias_table"),
    (rf"^{_B}blocks\.(\d+)\.mlp\.fc([12])\.(weight|bias)$", r"backbone.beit.layers.\1.mlp.fc\2.\3",
     r"^backbone\.beit\.layers\.(\d+)\.mlp\.fc([12])\.(weight|bias)$", r"core.core.pretrained.model.blocks.\1.mlp.fc\2.\3"),
    # DPT reassemble: act_postprocess{1..4} = [readout project, transpose, unflatten, 1x1 projection, resize]
    (r"^core\.core\.pretrained\.act_postprocess(\d)\.0\.project\.0\.(weight|bias)$", r"neck.reassemble_stage.readout_projects.{\1-1}.0.\2",
     r"^neck\.reassemble_stage\.readout_projects\.(\d)\.0\.(weight|bias)$", r"core.core.pretrained.act_postprocess{\1+1}.0.project.0.\2"),
    (r"^core\.core\.pretrained\.act_postprocess(\d)\.3\.(weight|bias)$", r"neck.reassemble_stage.layers.{\1-1}.projection.\2",
     r"^neck\.reassemble_stage\.layers\.(\d)\.projection\.(weight|bias)$", r"core.core.pretrained.act_postprocess{\1+1}.3.\2"),
    (r"^core\.core\.pretrained\.act_postprocess(\d)\.4\.(weight|bias)$", r"neck.reassemble_stage.layers.{\1-1}.resize.\2",
     r"^neck\.reassemble_stage\.layers\.(\d)\.resize\.(weight|bias)$", r"core.core.pretrained.act_postprocess{\1+1}.4.\2"),
    (r"^core\.core\.scratch\.layer(\d)_rn\.weight$", r"neck.convs.{\1-1}.weight",
     r"^neck\.convs\.(\d)\.weight$", r"core.core.scratch.layer{\1+1}_rn.weight"),
    # fusion: refinenet4 is applied first (HF fusion_stage.layers.0)
    (r"^core\.core\.scratch\.refinenet(\d)\.out_conv\.(weight|bias)$", r"neck.fusion_stage.layers.{4-\1}.projection.\2",
     r"^neck\.fusion_stage\.layers\.(\d)\.projection\.(weight|bias)$", r"core.core.scratch.refinenet{4-\1}.out_conv.\2"),
    (r"^core\.core\.scratch\.refinenet(\d)\.resConfUnit([12])\.conv([12])\.(weight|bias)$",
     r"neck.fusion_stage.layers.{4-\1}.residual_layer\2.convolution\3.\4",
     r"^neck\.fusion_stage\.layers\.(\d)\.residual_layer([12])\.convolution([12])\.(weight|bias)$",
     r"core.core.scratch.refinenet{4-\1}.resConfUnit\2.conv\3.\4"),
    # relative head: scratch.output_conv = [conv, upsample, conv, relu, conv, relu, identity]
    (r"^core\.core\.scratch\.output_conv\.0\.(weight|bias)$", r"relative_head.conv1.\1", r"^relative_head\.conv1\.(weight|bias)$", r"core.core.scratch.output_conv.0.\1"),
    (r"^core\.core\.scratch\.output_conv\.2\.(weight|bias)$", r"relative_head.conv2.\1", r"^relative_head\.conv2\.(weight|bias)$", r"core.core.scratch.output_conv.2.\1"),
    (r"^core\.core\.scratch\.output_conv\.4\.(weight|bias)$", r"relative_head.conv3.\1", r"^relative_head\.conv3\.(weight|bias)$", r"core.core.scratch.output_conv.4.\1"),
    # metric head
    (r"^conv2\.(weight|bias)$", r"metric_head.conv2.\1", r"^metric_head\.conv2\.(weight|bias)$", r"conv2.\1"),
    (r"^patch_transformer\.embedding_convPxP\.(weight|bias)$", r"metric_head.patch_transformer.embedding_convPxP.\1",
     r"^metric_head\.patch_transformer\.embedding_convPxP\.(weight|bias)$", r"patch_transformer.embedding_convPxP.\1"),
    (r"^patch_transformer\.transformer_encoder\.layers\.(\d+)\.(self_attn\.out_proj|linear1|linear2|norm1|norm2)\.(weight|bias)$",
     r"metric_head.patch_transformer.transformer_encoder.\1.\2.\3",
     r"^metric_head\.patch_transformer\.transformer_encoder\.(\d+)\.(self_attn\.out_proj|linear1|linear2|norm1|norm2)\.(weight|bias)$",
     r"patch_transformer.transformer_encoder.layers.\1.\2.\3"),
    (r"^mlp_classifier\.0\.(weight|bias)$", r"metric_head.mlp_classifier.linear1.\1", r"^metric_head\.mlp_classifier\.linear1\.(weight|bias)$", r"mlp_classifier.0.\1"),
    (r"^mlp_classifier\.2\.(weight|bias)$", r"metric_head.mlp_classifier.linear2.\1", r"^metric_head\.mlp_classifier\.linear2\.(weight|bias)$", r"mlp_classifier.2.\1"),
    # seed regressor(s), projectors, attractors: upstream Sequential "_net" = [conv, relu, conv(, act)]
    (r"^(seed_bin_regressors\.\w+|seed_bin_regressor|seed_projector|projectors\.\d+|attractors\.\w+\.\d+|attractors\.\d+)\._net\.0\.(weight|bias)$",
     r"metric_head.\1.conv1.\2",
     r"^metric_head\.(seed_bin_regressors\.\w+|seed_bin_regressor|seed_projector|projectors\.\d+|attractors\.\w+\.\d+|attractors\.\d+)\.conv1\.(weight|bias)$",
     r"\1._net.0.\2"),
    (r"^(seed_bin_regressors\.\w+|seed_bin_regressor|seed_projector|projectors\.\d+|attractors\.\w+\.\d+|attractors\.\d+)\._net\.2\.(weight|bias)$",
     r"metric_head.\1.conv2.\2",
     r"^metric_head\.(seed_bin_regressors\.\w+|seed_bin_regressor|seed_projector|projectors\.\d+|attractors\.\w+\.\d+|attractors\.\d+)\.conv2\.(weight|bias)$",
     r"\1._net.2.\2"),
    (r"^(conditional_log_binomial(?:\.\w+)?\.mlp\.[02])\.(weight|bias)$", r"metric_head.\1.\2",
     r"^metric_head\.(conditional_log_binomial(?:\.\w+)?\.mlp\.[02])\.(weight|bias)$", r"\1.\2"),
]
# buffers / modules of the upstream graph that the forward does not read
_UPSTREAM_IGNORED = (r"\.relative_position_index$", r"^core\.core\.pretrained\.model\.(norm|fc_norm|head|pos_embed|rel_pos_bias)\b",
                     r"^core\.prep\.", r"num_batches_tracked$")


def _expand(template: str, m: re.Match) -> str:
    """re.Match.expand plus {\\N+1} / {\\N-1} / {4-\\N} index arithmetic."""
    def arith(mm):
        expr = mm.group(1)
        expr = re.sub(r"\\(\d)", lambda g: m.group(int(g.group(1))), expr)
        return str(eval(expr, {"__builtins__": {}}))       # digits and + - only (the templates above)
    return m.expand(re.sub(r"\{([^}]*)\}", arith, template))


def upstream_to_hf(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """isl-org/ZoeDepth state dict (ZoeD_M12_NK.pt / _N / _K, the dict under "model") -> the names used here."""
    out, unknown = {}, []
    for k, v in sd.items():
        if any(re.search(p, k) for p in _UPSTREAM_IGNORED):
            continue
        m = re.match(rf"^{_B}blocks\.(\d+)\.attn\.qkv\.weight$", k)
        if m:       # timm Beit: fused [q; k; v] rows, biases only for q and v (modeling_beit.py:305-307)
            h = v.shape[0] // 3
            p = f"backbone.beit.layers.{m.group(1)}.attention."
            out[p + "q_proj.weight"], out[p + "k_proj.weight"], out[p + "v_proj.weight"] = v[:h], v[h:2 * h], v[2 * h:]
            continue
        m = re.match(rf"^{_B}blocks\.(\d+)\.attn\.(q|v)_bias$", k)
        if m:
            out[f"backbone.beit.layers.{m.group(1)}.attention.{m.group(2)}_proj.bias"] = v
            continue
        m = re.match(r"^patch_transformer\.transformer_encoder\.layers\.(\d+)\.self_attn\.in_proj_(weight|bias)$", k)
        if m:       # nn.MultiheadAttention's fused projection
            h = v.shape[0] // 3
            p = f"metric_head.patch_transformer.transformer_encoder.{m.group(1)}.self_attn."
            for j, n in enumerate(("query", "key", "value")):
                out[p + f"{n}.{m.group(2)}"] = v[j * h:(j + 1) * h]
            continue
        for up_re, hf_t, _, _ in _RULES:
            m = re.match(up_re, k)
            if m:
                out[_expand(hf_t, m)] = v
                break
        else:
            unknown.append(k)
    if unknown:
        raise KeyError(f"upstream ZoeDepth checkpoint: {len(unknown)} tensor names have no mapping, e.g. {unknown[:6]}")
    return out


def hf_to_upstream(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Inverse of upstream_to_hf (tests; exporting weights for the reference)."""
    out, unknown = {}, []
    qkv: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in sd.items():
        m = re.match(r"^backbone\.beit\.layers\.(\d+)\.attention\.(q|k|v)_proj\.(weight|bias)$", k)
        if m:
            i, n, wb = m.groups()
            if wb == "weight":
                qkv.setdefault(f"core.core.pretrained.model.blocks.{i}.attn.qkv.weight", {})[n] = v
            else:
                out[f"core.core.pretrained.model.blocks.{i}.attn.{n}_bias"] = v
            continue
        m = re.match(r"^metric_head\.patch_transformer\.transformer_encoder\.(\d+)\.self_attn\.(query|key|value)\.(weight|bias)$", k)
        if m:
            i, n, wb = m.groups()
            qkv.setdefault(f"patch_transformer.transformer_encoder.layers.{i}.self_attn.in_proj_{wb}", {})[n[0]] = v
            continue
        for _, _, hf_re, up_t in _RULES:
            m = re.match(hf_re, k)
            if m:
                out[_expand(up_t, m)] = v
                break
        else:
            unknown.append(k)
    for k, parts in qkv.items():
        out[k] = torch.cat([parts["q"], parts["k"], parts["v"]], 0)
    if unknown:
        raise KeyError(f"no upstream name for {unknown[:6]}")
    return out


def normalize_zoedepth_names(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Detect the container's naming scheme and return the names the engine reads."""
    if "model" in sd and isinstance(sd["model"], dict):          # upstream .pt: {"model": state_dict, (optimizer, epoch)}
        sd = sd["model"]
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    keys = list(sd.keys())
    if any(k.startswith("core.core.") for k in keys):
        return upstream_to_hf(sd)
    if any(k.startswith("backbone.encoder.layer.") for k in keys):
        return hf4_to_hf5(sd)
    return {k: v for k, v in sd.items() if not k.endswith("relative_position_index")}


def load_zoedepth_weights(path: str | None = None) -> Dict[str, torch.Tensor]:
    path = path or os.environ.get(ENV_ZOE)
    if not path:
        raise FileNotFoundError(
            f"no ZoeDepth weights: pass a path or set {ENV_ZOE} to a ZoeDepth checkpoint (Intel/zoedepth-nyu-kitti .safetensors, "
            "a .pt with HF names, or upstream ZoeD_M12_NK.pt); the reference's torch.hub download is not available offline")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return normalize_zoedepth_names(load_file(path))
    return normalize_zoedepth_names(torch.load(path, map_location="cpu", weights_only=True))


def load_cyclepose_checkpoint(path: str | None = None) -> Dict[str, torch.Tensor]:
    path = path or os.environ.get(ENV_POSE)
    if not path:
        raise FileNotFoundError(f"no CyclePose checkpoint: pass a path or set {ENV_POSE}")
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}
