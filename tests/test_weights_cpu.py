"""N1 weight ingestion (CPU): the three ZoeDepth containers load_zoedepth_weights accepts end up with the parameter names the
engine reads.  No upstream / hub checkpoint is reachable offline, so the name rules are exercised by round trips of synthetic
state dicts (HF-named -> inverse-mapped -> file -> loader -> identical tensors) plus spot checks of the names HF's
convert_zoedepth_to_hf.py documents."""
import dataclasses

import pytest
import torch

from bodyslam_amd import weights as W
from oracle import zoedepth_ref as Z


def _tiny(single=False, add_projection=True):
    cfg = dataclasses.replace(Z.tiny_config(), add_projection=add_projection)
    if single:
        cfg = dataclasses.replace(cfg, head_names=("nyu",))
    return cfg, Z.synth_weights(cfg, seed=3)


def test_upstream_has_no_relative_head_projection():
    """upstream MiDaS' scratch.output_conv starts with the 256 -> 128 conv: HF's optional relative_head.projection
    (config.add_projection) has no upstream counterpart, so an upstream checkpoint always yields the engine's no-projection graph"""
    _, sd = _tiny(False, True)
    with pytest.raises(KeyError, match="relative_head.projection"):
        W.hf_to_upstream(sd)


@pytest.mark.parametrize("single,add_projection", [(False, False), (True, False)])
def test_upstream_checkpoint_round_trip(tmp_path, single, add_projection):
    cfg, sd = _tiny(single, add_projection)
    up = W.hf_to_upstream(sd)
    # upstream naming: MiDaS core + ZoeDepth heads, fused q/k/v
    assert all(k.startswith(("core.core.", "conv2.", "patch_transformer.", "mlp_classifier.", "seed_", "projectors.", "attractors.",
                             "conditional_log_binomial.")) for k in up), sorted(up)[:5]
    assert up["core.core.pretrained.model.blocks.0.attn.qkv.weight"].shape == (3 * cfg.hidden, cfg.hidden)
    assert "core.core.pretrained.model.blocks.0.attn.q_bias" in up and "core.core.pretrained.model.blocks.0.attn.v_bias" in up
    assert "core.core.pretrained.model.blocks.0.attn.k_bias" not in up          # timm Beit has no key bias
    # the correspondences HF's conversion script documents
    pairs = {
        "core.core.scratch.refinenet4.resConfUnit1.conv1.weight": "neck.fusion_stage.layers.0.residual_layer1.convolution1.weight",
        "core.core.scratch.refinenet1.out_conv.bias": "neck.fusion_stage.layers.3.projection.bias",
        "core.core.pretrained.act_postprocess1.4.weight": "neck.reassemble_stage.layers.0.resize.weight",
        "core.core.pretrained.act_postprocess4.0.project.0.weight": "neck.reassemble_stage.readout_projects.3.0.weight",
        "core.core.scratch.layer3_rn.weight": "neck.convs.2.weight",
        "core.core.scratch.output_conv.2.weight": "relative_head.conv2.weight",
        "core.core.pretrained.model.blocks.1.gamma_2": "backbone.beit.layers.1.lambda_2",
        "seed_projector._net.2.bias": "metric_head.seed_projector.conv2.bias",
    }
    if not single:
        pairs["seed_bin_regressors.kitti._net.0.weight"] = "metric_head.seed_bin_regressors.kitti.conv1.weight"
        pairs["attractors.nyu.3._net.2.weight"] = "metric_head.attractors.nyu.3.conv2.weight"
        pairs["patch_transformer.transformer_encoder.layers.2.linear1.weight"] = "metric_head.patch_transformer.transformer_encoder.2.linear1.weight"
        assert up["patch_transformer.transformer_encoder.layers.0.self_attn.in_proj_weight"].shape == (3 * cfg.pt_hidden, cfg.pt_hidden)
    else:
        pairs["seed_bin_regressor._net.0.weight"] = "metric_head.seed_bin_regressor.conv1.weight"
        pairs["attractors.2._net.0.bias"] = "metric_head.attractors.2.conv1.bias"
        pairs["conditional_log_binomial.mlp.2.weight"] = "metric_head.conditional_log_binomial.mlp.2.weight"
    for u, h in pairs.items():
        assert u in up and torch.equal(up[u], sd[h]), (u, h)
    # the file the reference's torch.hub call downloads: {"model": state_dict, ...}, with buffers the forward does not read
    up["core.core.pretrained.model.blocks.0.attn.relative_position_index"] = torch.zeros(5, 5, dtype=torch.long)
    path = str(tmp_path / "ZoeD_M12_NK.pt")
    torch.save({"model": up, "epoch": 0}, path)
    back = W.load_zoedepth_weights(path)
    assert set(back) == set(sd)
    assert all(torch.equal(back[k], sd[k]) for k in sd)


def test_hub_checkpoint_names_are_renamed(tmp_path):
    """Intel/zoedepth-nyu-kitti on the hub carries transformers 4.x names; the rules of transformers 5's conversion_mapping
    ("ViTModel" + "BeitModel" + "BeitBackbone") are applied."""
    from safetensors.torch import save_file
    cfg, sd = _tiny(False, False)
    inv = [(".attention.q_proj.", ".attention.attention.query."), (".attention.k_proj.", ".attention.attention.key."),
           (".attention.v_proj.", ".attention.attention.value."), (".attention.o_proj.", ".attention.output.dense."),
           (".mlp.fc1.", ".intermediate.dense."), (".mlp.fc2.", ".output.dense."),
           (".relative_position_bias.relative_position_bias_table", ".attention.attention.relative_position_bias.relative_position_bias_table"),
           ("backbone.beit.layers.", "backbone.encoder.layer."), ("backbone.beit.embeddings.", "backbone.embeddings.")]
    old = {}
    for k, v in sd.items():
        if k.startswith("backbone."):
            for a, b in inv:
                k = k.replace(a, b)
        old[k] = v.contiguous()
    assert "backbone.encoder.layer.0.attention.attention.query.weight" in old and "backbone.encoder.layer.0.output.dense.bias" in old
    old["backbone.encoder.layer.0.attention.attention.relative_position_bias.relative_position_index"] = torch.zeros(3, 3, dtype=torch.long)
    path = str(tmp_path / "model.safetensors")
    save_file(old, path)
    back = W.load_zoedepth_weights(path)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    # names already in the engine's scheme pass through unchanged
    same = W.normalize_zoedepth_names(dict(sd))
    assert set(same) == set(sd)


def test_unknown_upstream_tensor_fails_loudly():
    _, sd = _tiny(False, False)
    up = W.hf_to_upstream(sd)
    up["core.core.scratch.something_new.weight"] = torch.zeros(1)
    with pytest.raises(KeyError, match="something_new"):
        W.upstream_to_hf(up)


def test_loaded_names_are_what_hf_builds():
    """the normalised names are exactly the state-dict names of the installed HF ZoeDepthForDepthEstimation"""
    tr = pytest.importorskip("transformers")
    from oracle.make_golden import hf_config
    cfg, sd = _tiny(False, False)
    m = tr.ZoeDepthForDepthEstimation(hf_config(cfg))
    hf_names = {k for k in m.state_dict().keys() if not k.endswith("relative_position_index")}
    assert set(W.upstream_to_hf(W.hf_to_upstream(sd))) == hf_names


def test_neck_mode_forms():
    """which neck / head products run weight-only under the forms of ZoeDepthEngine.neck_mode (the per-site calibration writes "wonly:...")"""
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    e = ZoeDepthEngine.__new__(ZoeDepthEngine)
    for mode, expect in (("full", (False, False, False)), ("w", (True, True, False)), ("ro,ra,nc,fu,pj,mh", (False, True, False)),
                         ("wonly:fu3.r1.c1.w,rh.conv1.w", (True, False, False)), ("wonly:rh.projection.w", (False, True, False))):
        e.neck_mode = mode
        got = (e.neck_site_wonly("fu3.r1.c1.w"), e.neck_site_wonly("rh.projection.w"), e.neck_site_wonly("ro2.w_cls"))
        assert got == expect, (mode, got)
        assert not e.neck_site_plain("fu3.r1.c1.w") and not e.neck_site_plain("rh.projection.w")
    # the second stage's form: one-pass sites are a subset of the weight-only ones (their producers skip the lo8 plane all the same); the fused
    # up-convolution has no one-pass form and the per-image readout bias is never touched
    e.neck_mode = "wonly:fu3.r1.c1.w,rh.conv1.w,rh.conv2.w,rh.projection.w;plain:rh.conv1.w,rh.conv2.w,rh.projection.w"
    assert [e.neck_site_wonly(k) for k in ("fu3.r1.c1.w", "rh.conv1.w", "rh.conv2.w", "rh.projection.w", "nc0.w", "ro2.w_cls")] == [True, True, True, True, False, False]
    assert [e.neck_site_plain(k) for k in ("fu3.r1.c1.w", "rh.conv1.w", "rh.conv2.w", "rh.projection.w", "nc0.w", "ro2.w_cls")] == [False, True, False, True, False, False]
