"""Pins oracle/zoedepth_ref.py against (a) committed golden vectors produced with HF
ZoeDepthForDepthEstimation (oracle/make_golden.py) and (b) HF live, when transformers is importable
(it is part of this image).  HF transformers 5.15.0 is the installed weight-compatible restatement
of the un-vendored upstream isl-org/ZoeDepth the reference pulls through torch.hub
(BodySLAM_Refactored/src/depth_estimation/interface.py:46)."""
import os

import numpy as np
import pytest
import torch

from oracle import zoedepth_ref as Z


@pytest.fixture(scope="module")
def tiny(golden_dir):
    return np.load(os.path.join(golden_dir, "zoedepth_tiny.npz"))


@pytest.mark.parametrize("tag,rb,route", [("nyu", 3.0, 0), ("kitti", -3.0, 1)])
def test_tiny_matches_hf_golden(tiny, tag, rb, route):
    cfg = Z.tiny_config()
    w = Z.synth_weights(cfg, seed=int(tiny["weight_seed"]), route_bias=rb)
    rng = np.random.default_rng(int(tiny["input_seed"]))
    x = torch.from_numpy(rng.standard_normal((1, 3, 64, 96), dtype=np.float32))
    taps = {}
    with torch.no_grad():
        d, lg = Z.zoedepth_forward(w, cfg, x, taps)
    assert int(taps["route"][0]) == route
    assert np.allclose(lg.numpy(), tiny[f"logits_{tag}"], atol=1e-5)
    assert np.abs(d.numpy() - tiny[f"depth_{tag}"]).max() < 2e-5
    assert d.min() > 0.05 and d.max() < 10 and (d.max() - d.min()) > 0.5  # non-degenerate output


def test_per_image_route_is_batch_invariant():
    """The reference always runs batch 1 (interface.py:61), so image i's result must not depend on
    what else is in the batch (HF's batch-summed vote, modeling_zoedepth.py:1063-1067, would)."""
    cfg = Z.tiny_config()
    w = Z.synth_weights(cfg, seed=1)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((3, 3, 64, 96), dtype=np.float32))
    with torch.no_grad():
        d_all, _ = Z.zoedepth_forward(w, cfg, x)
        d_one = torch.cat([Z.zoedepth_forward(w, cfg, x[i:i + 1])[0] for i in range(3)])
    assert np.abs(d_all.numpy() - d_one.numpy()).max() < 2e-5


@pytest.mark.parametrize("add_projection", [True, False])
def test_live_against_hf_tiny(add_projection):
    """add_projection False = HF's config default (no conv in front of the relative head): the oracle keys on the weight"""
    import dataclasses
    tr = pytest.importorskip("transformers")
    from oracle.make_golden import hf_config
    cfg = dataclasses.replace(Z.tiny_config(), add_projection=add_projection)
    w = Z.synth_weights(cfg, seed=9)
    m = tr.ZoeDepthForDepthEstimation(hf_config(cfg)).eval()
    m.load_state_dict(w, strict=True)
    assert ("relative_head.projection.weight" in w) == add_projection
    x = torch.randn(2, 3, 96, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        o = m(pixel_values=x)
        d, lg = Z.zoedepth_forward(w, cfg, x, per_image_route=False)
    assert np.abs(o.predicted_depth.numpy() - d.numpy()).max() < 2e-5
    assert np.abs(o.domain_logits.numpy() - lg.numpy()).max() < 1e-5


@pytest.mark.parametrize("name", ["nyu", "kitti"])
def test_live_against_hf_tiny_single_head(name):
    """ZoeD_N / ZoeD_K: one bin configuration -> HF builds ZoeDepthMetricDepthEstimationHead (no router, per-level attractor
    counts 16/8/4/1, the relative depth as a 33rd input of the log-binomial MLP).  strict=True pins the parameter names."""
    import dataclasses
    tr = pytest.importorskip("transformers")
    from oracle.make_golden import hf_config
    cfg = dataclasses.replace(Z.tiny_config(), head_names=(name,))
    w = Z.synth_weights(cfg, seed=11)
    m = tr.ZoeDepthForDepthEstimation(hf_config(cfg)).eval()
    m.load_state_dict(w, strict=True)
    x = torch.randn(2, 3, 96, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        o = m(pixel_values=x)
        d, _ = Z.zoedepth_forward(w, cfg, x)
    assert o.domain_logits is None
    assert np.abs(o.predicted_depth.numpy() - d.numpy()).max() < 2e-5


def test_pre_post_geometry():
    assert Z.pad_sizes(480, 640) == (46, 53)
    assert Z.net_size(480 + 92, 640 + 106) == (384, 512)
    assert Z.pad_sizes(480, 600) == (46, 51)
    assert Z.net_size(480 + 92, 600 + 102) == (416, 512)
    assert Z.net_size(1024 + 2 * 67, 1280 + 2 * 75) == (416, 512)
    f = torch.randint(0, 256, (1, 480, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    x = Z.preprocess(f)
    assert x.shape == (1, 3, 384, 512) and x.min() >= -1 and x.max() <= 1
    d = torch.rand(1, 384, 512) + 1
    out = Z.postprocess(d, torch.flip(d, dims=[-1]), 480, 640)
    assert out.shape == (1, 480, 640)
    u16 = Z.to_uint16(out)
    assert u16.dtype == np.uint16 and u16.min() >= 128  # bicubic may undershoot random data a little


def test_preprocess_matches_hf_processor():
    """pad + resize + normalise against HF's ZoeDepthImageProcessorPil (cites upstream depth_model.py#L57)."""
    pytest.importorskip("transformers")
    from transformers.models.zoedepth.image_processing_pil_zoedepth import ZoeDepthImageProcessorPil
    # upstream PrepForMidas uses ensure_multiple_of=32 (as does the released Intel/zoedepth-nyu-kitti
    # preprocessor config); the bare class default (1/32) is not what any checkpoint ships with
    proc = ZoeDepthImageProcessorPil(ensure_multiple_of=32)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    ref = proc(images=[img], return_tensors="pt")["pixel_values"]
    x = Z.preprocess(torch.from_numpy(img)[None])
    assert ref.shape == x.shape
    assert (ref - x).abs().max() < 1e-5


@pytest.mark.slow
def test_full_size_matches_hf_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "zoedepth_full.npz"))
    cfg = Z.ZOED_NK
    w = Z.synth_weights(cfg, seed=int(g["weight_seed"]))
    rng = np.random.default_rng(int(g["input_seed"]))
    x = torch.from_numpy(rng.standard_normal((1, 3, 384, 512), dtype=np.float32))
    with torch.no_grad():
        d, lg = Z.zoedepth_forward(w, cfg, x)
    assert np.abs(d.numpy()[:, ::8, ::8] - g["depth_sub"]).max() < 5e-5
    assert abs(d.mean().item() - float(g["depth_mean"])) < 1e-5
    assert np.allclose(lg.numpy(), g["logits"], atol=1e-4)
