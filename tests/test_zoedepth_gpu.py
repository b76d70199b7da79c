"""MDEM parity: the HIP ZoeDepth engine (through the C ABI) against the CPU oracle
(oracle/zoedepth_ref.py, pinned against HF ZoeDepth) on identical seeded weights and frames.
Needs an MI355X: `pytest -m gpu`."""
import dataclasses
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "zoedepth_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    print(line)


def small_oracle_cfg():
    from oracle import zoedepth_ref as Z
    # head_dim must stay 64 (the BEiT-L value the attention kernel is written for)
    return Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)


def product_cfg(ocfg):
    from bodyslam_amd.zoedepth import ZoeConfig
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    return ZoeConfig(**{k: v for k, v in dataclasses.asdict(ocfg).items() if k in names})


def to_nchw(t, meta):
    kind = meta[0]
    t_raw = t
    t = t.float().cpu()
    if kind == "tokens":
        return t.view(meta[1], meta[2], meta[3])
    if kind == "tokens_grouped":     # rows: the NB cls tokens, then the S-1 patch tokens of every image -> [NB, S, hidden], cls first
        nb, S, hd, cp = meta[1], meta[2], meta[3], meta[4]      # cp: first patch row (the cls group is padded to whole GEMM tiles)
        return torch.cat([t[:nb].view(nb, 1, hd), t[cp: cp + nb * (S - 1)].view(nb, S - 1, hd)], 1)
    if kind == "nhwc" and len(meta) > 5 and meta[5] == 3:
        # (hi16 | hi8 | -): a map whose lo8 plane was not written (every consumer is weight-only): the value is hi16
        C = meta[4]
        return t_raw.cpu().view(meta[1], meta[2], meta[3], 2 * C)[..., :C].float().permute(0, 3, 1, 2)
    if kind == "nhwc" and len(meta) > 5 and meta[5] == 2:
        # accurate mode, FP8 pair format: (hi16 | hi8 | lo8) per pixel; the value is hi16 + lo8 * 2^-LO_EXP
        from bodyslam_amd import _lib as L
        C = meta[4]
        raw = t_raw.cpu().view(meta[1], meta[2], meta[3], 2 * C)
        hi = raw[..., :C].float()
        planes = raw[..., C:].contiguous().view(torch.uint8).view(meta[1], meta[2], meta[3], 2 * C)
        lo = planes[..., C:].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -L.F8_ACT_LO_EXP
        return (hi + lo).permute(0, 3, 1, 2)
    if kind == "nhwc" and len(meta) > 5 and meta[5]:
        # accurate mode: (hi | lo) channel pairs; the value is their sum
        t = t.view(meta[1], meta[2], meta[3], 2, meta[4])
        return (t[..., 0, :] + t[..., 1, :]).permute(0, 3, 1, 2)
    if kind in ("nhwc", "nhwc_route"):
        return t.view(meta[1], meta[2], meta[3], meta[4]).permute(0, 3, 1, 2)
    return t


# Asserted per-tap tolerances: max |error| of an intermediate over max(|reference|, 1), by (precision, storage type).  About 4x the
# worst value seen over every configuration of this file on MI355X (accurate fp16 3.5e-4 / bins 1.2e-3, fast fp16 1.8e-3 / 3.8e-3,
# fast bf16 1.5e-2 / 8.4e-3): a regression in one stage fails here even when the log-binomial head happens to absorb it in the
# final depth L1.
TAP_TOL = {("accurate", "f16"): (1.5e-3, 5e-3), ("accurate", "bf16"): (2e-2, 2e-2), ("fast", "f16"): (7e-3, 1.5e-2), ("fast", "bf16"): (6e-2, 4e-2)}


def compare_taps(taps_p, taps_o, route_o, tag, heads=("nyu", "kitti"), tol=None):
    """per-tap max errors (reported); with tol = (precision, dtype name) every tap is ASSERTED against TAP_TOL"""
    worst = {}
    for name, (t, meta) in taps_p.items():
        if name in ("logits",):
            continue
        got = to_nchw(t, meta)
        if meta[0] == "nhwc_route":
            # both heads' bins are carried side by side; only the routed head is defined per image
            errs = []
            for hn, hname in enumerate(heads):      # slot order = the configuration's head order
                key = f"{hname}.{name}"
                if key not in taps_o:
                    continue
                sel, ref = taps_o[key]
                for j, b in enumerate(sel.tolist()):
                    errs.append((got[b, hn * 64:(hn + 1) * 64] - ref[j]).abs().max().item())
            e = max(errs)
            scale = 1.0
        else:
            if name not in taps_o:
                continue
            ref = taps_o[name]
            if name.startswith("layer") or name == "embed":
                ref = ref
            e = (got - ref).abs().max().item()
            scale = ref.abs().max().item()
        worst[name] = (e, scale)
        report(f"  [{tag}] {name:14s} max|err|={e:.3e}  (ref max {scale:.3e}, rel {e / max(scale, 1e-9):.2e})")
        if tol is not None:
            lim = TAP_TOL[tol][1 if name.startswith("bins") else 0]
            assert e / max(scale, 1.0) <= lim, f"[{tag}] tap {name}: max|err| {e:.3e} over scale {scale:.3e} exceeds {lim:.1e}"
    return worst


_ORACLE_CACHE = {}


def oracle_case(cfg_o, B, H, W, target_hw, seed, route_bias, flip, weights_hook=None):
    """Oracle side of a case (weights, frames, taps, final depth); shared by the dtype variants.  weights_hook(w) edits the seeded
    weights in place (adversarial statistics) and must have a __name__."""
    from bodyslam_amd.synthetic import make_sequence
    from oracle import zoedepth_ref as Z
    key = (repr(cfg_o), B, H, W, target_hw, seed, route_bias, flip, getattr(weights_hook, "__name__", None))
    if key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    w = Z.synth_weights(cfg_o, seed=seed, route_bias=route_bias)
    if weights_hook is not None:
        weights_hook(w)
    frames = torch.from_numpy(make_sequence(B, H, W, seed=seed))
    with torch.no_grad():
        x = Z.preprocess(frames, target_hw)
        xin = torch.cat([x, torch.flip(x, dims=[3])], 0) if flip else x
        taps_o = {}
        t0 = time.time()
        d, logits = Z.zoedepth_forward(w, cfg_o, xin, taps_o)
        t_or = time.time() - t0
        ref = Z.postprocess(d[:B], d[B:] if flip else None, H, W)
    _ORACLE_CACHE.clear()          # keep at most one case resident (full-size weights are 1.4 GB)
    _ORACLE_CACHE[key] = (w, frames, taps_o, logits, ref, t_or)
    return _ORACLE_CACHE[key]


def run_case(cfg_o, dtype, B, H, W, target_hw, seed, route_bias=0.0, flip=True, precision="fast", weights_hook=None, **eng_kw):
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    from oracle import zoedepth_ref as Z
    w, frames, taps_o, logits, ref, t_or = oracle_case(cfg_o, B, H, W, target_hw, seed, route_bias, flip, weights_hook)
    eng = ZoeDepthEngine(w, product_cfg(cfg_o), dtype=dtype, target_hw=target_hw, precision=precision, **eng_kw)
    taps_p = {}
    dm, du = eng.infer(frames.cuda(), flip_aug=flip, taps=taps_p)
    torch.cuda.synchronize()
    lp = taps_p["logits"][0].cpu()[:, :2] if "logits" in taps_p else None        # single-head models have no router
    return dict(dm=dm.cpu(), du=du.cpu().numpy().view(np.uint16), ref=ref, taps_p=taps_p, taps_o=taps_o, logits_o=logits,
                logits_p=lp, route_p=eng.plan_for(B, H, W, flip).route.cpu(), t_oracle=t_or, Z=Z, calibration=eng.calibration, eng=eng)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("route_bias", [3.0, -3.0])
def test_small_backbone_full_head(dtype, route_bias):
    """Small BEiT (4 layers, hidden 128) with the full-size neck + heads at a 96x128 network input: every
    kernel and every epilogue mode of the forward runs; both metric heads are forced in turn."""
    r = run_case(small_oracle_cfg(), dtype, B=2, H=120, W=160, target_hw=(96, 128), seed=3, route_bias=route_bias)
    tag = f"small {str(dtype)[6:]} rb{route_bias:+.0f}"
    compare_taps(r["taps_p"], r["taps_o"], None, tag, tol=("fast", "f16" if dtype == torch.float16 else "bf16"))
    route_o = torch.argmax(r["logits_o"], -1)
    report(f"  [{tag}] logits oracle {r['logits_o'].tolist()} hip {r['logits_p'].tolist()}")
    assert torch.equal(route_o.int(), r["route_p"]) and (route_o == (0 if route_bias > 0 else 1)).all()
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    mx = (r["dm"] - r["ref"]).abs().max().item()
    report(f"[{tag}] depth L1={l1:.3e} max={mx:.3e} (depth range {r['ref'].min():.3f}..{r['ref'].max():.3f})")
    assert l1 < (2e-3 if dtype == torch.float16 else 2e-2)
    lsb = np.abs(r["du"].astype(np.int32) - r["Z"].to_uint16(r["ref"]).astype(np.int32))
    report(f"[{tag}] u16: max |diff| = {lsb.max()} LSB, mean {lsb.mean():.3f}")


@pytest.mark.parametrize("precision", ["accurate", "fast"])
def test_small_without_relative_head_projection(precision):
    """HF's config default add_projection=False (a checkpoint without relative_head.projection.*): the engine follows the weights
    and feeds the last fused map straight into relative_head.conv1."""
    cfg_o = dataclasses.replace(small_oracle_cfg(), add_projection=False)
    r = run_case(cfg_o, torch.float16, B=2, H=120, W=160, target_hw=(96, 128), seed=6, precision=precision)
    compare_taps(r["taps_p"], r["taps_o"], None, f"small no-projection {precision}", tol=(precision, "f16"))
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    report(f"[small no-projection {precision}] depth L1={l1:.3e}")
    assert torch.equal(torch.argmax(r["logits_o"], -1).int(), r["route_p"])
    assert l1 < (1e-4 if precision == "accurate" else 2e-3)


def test_hip_graph_replay_equals_eager():
    """Plan.capture(): the two-lane launch sequence as one HIP graph; replays give the eager result bit for bit, for new inputs too."""
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    from oracle import zoedepth_ref as Z
    cfg_o = small_oracle_cfg()
    eng = ZoeDepthEngine(Z.synth_weights(cfg_o, seed=4), product_cfg(cfg_o), dtype=torch.float16, target_hw=(96, 128), precision="accurate")
    frames = torch.from_numpy(make_sequence(4, 120, 160, seed=9)).cuda()
    eager = [eng.infer(frames[i:i + 2])[0].clone() for i in (0, 2)]
    g0 = eng.infer(frames[0:2], graph=True)[0].clone()
    assert eng.plan_for(2, 120, 160, True).plan._graph is not None
    g1 = eng.infer(frames[2:4], graph=True)[0].clone()
    g0b = eng.infer(frames[0:2], graph=True)[0].clone()
    assert torch.equal(g0, eager[0]) and torch.equal(g1, eager[1]) and torch.equal(g0b, eager[0])
    taps = {}
    eng.infer(frames[0:2], taps=taps)           # the tap path still runs eagerly on a captured plan
    assert "depth_net" in taps


def test_small_accurate_mode():
    """precision="accurate" (split-precision products in the neck / heads, split weights in the backbone) on the small case:
    every split code path (cast_split, relu_split, split resize / add_resized / logbinom, 2- and 3-segment GEMMs) runs and must
    beat the single-pass result by a wide margin."""
    kw = dict(B=2, H=120, W=160, target_hw=(96, 128), seed=3, route_bias=3.0)
    fast = run_case(small_oracle_cfg(), torch.float16, **kw)
    r = run_case(small_oracle_cfg(), torch.float16, precision="accurate", **kw)
    compare_taps(r["taps_p"], r["taps_o"], None, "small f16 accurate", tol=("accurate", "f16"))
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    l1_fast = (fast["dm"] - fast["ref"]).abs().mean().item()
    report(f"[small f16 accurate] depth L1={l1:.3e} (fast mode {l1_fast:.3e})")
    assert torch.equal(torch.argmax(r["logits_o"], -1).int(), r["route_p"])
    assert l1 < 0.5 * l1_fast and l1 < 5e-4


def test_batch_and_noflip_invariance():
    """Image i's depth must not depend on batch size or on the other images (per-image routing), and the
    no-flip path must equal the first half of the flip path's network output."""
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    from oracle import zoedepth_ref as Z
    cfg_o = small_oracle_cfg()
    w = Z.synth_weights(cfg_o, seed=4)
    eng = ZoeDepthEngine(w, product_cfg(cfg_o), dtype=torch.float16, target_hw=(96, 128))
    frames = torch.from_numpy(make_sequence(3, 120, 160, seed=9)).cuda()
    d3 = eng.infer(frames)[0].clone()
    d1 = torch.cat([eng.infer(frames[i:i + 1])[0].clone() for i in range(3)])
    assert torch.equal(d3, d1)
    n3 = eng.plan_for(3, 120, 160, True).depth_net.clone()
    eng.infer(frames, flip_aug=False)
    n3nf = eng.plan_for(3, 120, 160, False).depth_net.clone()
    assert torch.equal(n3[:3], n3nf)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_full_size_zoed_nk(dtype):
    """The real configuration: BEiT-L/16 (24 layers, hidden 1024), 640x480 frame -> 384x512 network input."""
    from oracle import zoedepth_ref as Z
    r = run_case(Z.ZOED_NK, dtype, B=1, H=480, W=640, target_hw=(384, 512), seed=1)
    tag = f"ZoeD_NK {str(dtype)[6:]}"
    compare_taps(r["taps_p"], r["taps_o"], None, tag, tol=("fast", "f16" if dtype == torch.float16 else "bf16"))
    report(f"  [{tag}] logits oracle {r['logits_o'].tolist()} hip {r['logits_p'].tolist()}  (oracle forward {r['t_oracle']:.1f}s)")
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    mx = (r["dm"] - r["ref"]).abs().max().item()
    report(f"[{tag}] 640x480 depth L1={l1:.3e} m, max={mx:.3e} m (depth range {r['ref'].min():.3f}..{r['ref'].max():.3f} m)")
    lsb = np.abs(r["du"].astype(np.int32) - Z.to_uint16(r["ref"]).astype(np.int32))
    report(f"[{tag}] u16: max |diff| = {lsb.max()} LSB, mean {lsb.mean():.3f} LSB")
    route_o = torch.argmax(r["logits_o"], -1)
    assert torch.equal(route_o.int(), r["route_p"])
    # stated tolerance (DESIGN.md "Numerics"): fp16 operands / fp32 accumulate through 24 layers
    assert l1 < (1e-3 if dtype == torch.float16 else 1e-2)


def test_full_size_zoed_nk_accurate():
    """precision="accurate" at the real configuration.  Tolerance: the north star's depth L1 <= 1e-4 m against the fp32
    oracle (DESIGN.md "Numerics"), which single-pass fp16 (3e-4) does not reach."""
    from oracle import zoedepth_ref as Z
    r = run_case(Z.ZOED_NK, torch.float16, B=1, H=480, W=640, target_hw=(384, 512), seed=1, precision="accurate")
    tag = "ZoeD_NK f16 accurate"
    compare_taps(r["taps_p"], r["taps_o"], None, tag, tol=("accurate", "f16"))
    report(f"[{tag}] calibration: {r['calibration']}")
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    mx = (r["dm"] - r["ref"]).abs().max().item()
    report(f"[{tag}] 640x480 depth L1={l1:.3e} m, max={mx:.3e} m, mean signed {(r['dm'] - r['ref']).mean().item():+.3e} m")
    lsb = np.abs(r["du"].astype(np.int32) - Z.to_uint16(r["ref"]).astype(np.int32))
    report(f"[{tag}] u16: max |diff| = {lsb.max()} LSB, mean {lsb.mean():.3f} LSB")
    assert torch.equal(torch.argmax(r["logits_o"], -1).int(), r["route_p"])
    assert l1 <= 1e-4


def test_full_size_zoed_nk_accurate_bf16():
    """accurate mode on bf16 storage: hi has 8 significant bits, the e4m3 correction planes add ~4: 2e-4 m measured (bf16
    single-pass: 4.8e-3).  Stated tolerance 5e-4 m; fp16 is the type that meets the north star's 1e-4."""
    from oracle import zoedepth_ref as Z
    r = run_case(Z.ZOED_NK, torch.bfloat16, B=1, H=480, W=640, target_hw=(384, 512), seed=1, precision="accurate")
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    report(f"[ZoeD_NK bf16 accurate] 640x480 depth L1={l1:.3e} m")
    assert torch.equal(torch.argmax(r["logits_o"], -1).int(), r["route_p"])
    assert l1 <= 5e-4


def test_bf16_reference_precision():
    """BASELINE config 2 names bf16.  With e4m3 correction planes bf16 storage carries 8 + 4 significant bits -- fp16's single pass -- and
    misses the tolerance (test_full_size_bf16_accurate: ~1.3e-4 m).  As (hi | lo) bf16 pairs, three MFMA passes per product (precision=
    "reference"), it carries 16: this is the bf16 configuration that meets 1e-4 m."""
    from oracle import zoedepth_ref as Z
    r = run_case(Z.ZOED_NK, torch.bfloat16, B=1, H=480, W=640, target_hw=(384, 512), seed=1, precision="reference")      # (seed 1: the oracle case of the tests above)
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    report(f"[ZoeD_NK bf16 reference precision (3-pass pairs)] 640x480 depth L1={l1:.3e} m, max={(r['dm'] - r['ref']).abs().max().item():.3e} m")
    assert l1 <= 1e-4


@pytest.mark.parametrize("H,W", [(480, 600), (1024, 1280)])
def test_other_frame_geometries(H, W):
    """BASELINE config 1 (the reference's own 600x480 example image) and config 5 (1280x1024): both resolve to a 416x512
    network input (833 tokens, rel-pos window 26x32) -- another attention tiling, other pads and resize ratios."""
    from oracle import zoedepth_ref as Z
    r = run_case(Z.ZOED_NK, torch.float16, B=1, H=H, W=W, target_hw=(384, 512), seed=2, precision="accurate")
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    mx = (r["dm"] - r["ref"]).abs().max().item()
    report(f"[ZoeD_NK f16 accurate {W}x{H}] depth L1={l1:.3e} m, max={mx:.3e} m")
    lsb = np.abs(r["du"].astype(np.int32) - Z.to_uint16(r["ref"]).astype(np.int32))
    assert torch.equal(torch.argmax(r["logits_o"], -1).int(), r["route_p"])
    assert l1 <= 1e-4 and lsb.max() <= 1


@pytest.mark.parametrize("name", ["nyu", "kitti"])
@pytest.mark.parametrize("precision", ["accurate", "fast"])
def test_single_head_models_small(name, precision):
    """ZoeD_N / ZoeD_K (one metric head, HF ZoeDepthMetricDepthEstimationHead): no router, attractor counts 16/8/4/1, hidden widths
    256 / 128 / 80, the relative depth as a 33rd input of the log-binomial MLP -- small backbone, full-size neck and head."""
    cfg_o = dataclasses.replace(small_oracle_cfg(), head_names=(name,))
    r = run_case(cfg_o, torch.float16, B=2, H=120, W=160, target_hw=(96, 128), seed=5, precision=precision)
    compare_taps(r["taps_p"], r["taps_o"], None, f"small {name} {precision}", heads=(name,), tol=(precision, "f16"))
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    report(f"[small single-head {name} {precision}] depth L1={l1:.3e} (range {r['ref'].min():.3f}..{r['ref'].max():.3f})")
    assert (r["route_p"] == 0).all()
    assert l1 < (1e-4 if precision == "accurate" else 2e-3)


def test_full_size_zoed_n_accurate():
    """ZoeD_N at the real size (BEiT-L, 640x480), accurate mode: the north star's 1e-4 m."""
    from oracle import zoedepth_ref as Z
    r = run_case(Z.ZOED_N, torch.float16, B=1, H=480, W=640, target_hw=(384, 512), seed=3, precision="accurate")
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    report(f"[ZoeD_N f16 accurate] 640x480 depth L1={l1:.3e} m, max={(r['dm'] - r['ref']).abs().max().item():.3e} m")
    assert l1 <= 1e-4


@pytest.mark.parametrize("hook", [None, "outlier"])
def test_reference_precision_engine(hook):
    """precision="reference" (three 16-bit passes on (hi | lo) pairs for every product, split-precision attention): the on-device stand-in
    for the fp32 oracle that calibrate() measures the production modes against.  It must sit an order of magnitude inside the 1e-4 m
    tolerance on plain AND on outlier-channel weights, or the absolute check would be worth nothing."""
    from oracle import zoedepth_ref as Z
    wh = _hook_outlier_channels if hook else None
    r = run_case(Z.ZOED_NK, torch.float16, B=1, H=480, W=640, target_hw=(384, 512), seed=9, precision="reference", weights_hook=wh)
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    mx = (r["dm"] - r["ref"]).abs().max().item()
    report(f"[ZoeD_NK f16 reference precision, weights hook {hook}] 640x480 depth L1={l1:.3e} m, max={mx:.3e} m")
    assert r["eng"].plan_for(1, 480, 640, True).attn_corr
    # (what is left on the outlier weights: the attractor MLPs of the bins head, single products in every mode, 1.1e-5 m;
    # tools/probes/outlier_rounding_study.py mh16)
    assert l1 <= (1e-5 if hook is None else 3e-5)


# ------------------------------------------------------------------------------------------------
# The tolerance against weights that do not look like the seeded ones: trained BEiT-L checkpoints carry a
# per-channel layer-scale spanning decades, a few outlier channels and heavy-tailed weights.  "auto" (the default) must hold the
# north star's 1e-4 m on each, whatever it has to switch back on; the fixed cheap mode is reported beside it.
# (the constructions live in bodyslam_amd/synthetic.py: bench.py --weights outlier measures the same set)
from bodyslam_amd.synthetic import heavy_tailed as _hook_heavy_tailed  # noqa: E402
from bodyslam_amd.synthetic import layerscale_wide as _hook_layerscale_wide  # noqa: E402
from bodyslam_amd.synthetic import outlier_channels as _hook_outlier_channels  # noqa: E402


@pytest.mark.parametrize("hook", [_hook_outlier_channels, _hook_layerscale_wide, _hook_heavy_tailed])      # (outlier first: the oracle case of the test above)
def test_auto_modes_hold_tolerance_on_adversarial_weights(hook):
    from oracle import zoedepth_ref as Z
    kw = dict(B=1, H=480, W=640, target_hw=(384, 512), seed=9, precision="accurate", weights_hook=hook)
    r = run_case(Z.ZOED_NK, torch.float16, **kw)                       # class_modes = "auto" (default)
    l1 = (r["dm"] - r["ref"]).abs().mean().item()
    cal = r["calibration"]
    del r
    torch.cuda.empty_cache()
    rw = run_case(Z.ZOED_NK, torch.float16, class_modes="wmean", neck_mode="full", **kw)
    l1w = (rw["dm"] - rw["ref"]).abs().mean().item()
    report(f"[adversarial {hook.__name__}] auto: L1={l1:.3e} m with {cal['class_modes']} neck={cal['neck_mode']!r} "
           f"(vs all-full {cal['l1_total_vs_full_m']:.2e}; per class {({k: round(v, 7) for k, v in cal['l1_vs_full_m'].items()})}); "
           f"fixed wmean: L1={l1w:.3e} m; depth range {rw['ref'].min():.3f}..{rw['ref'].max():.3f}")
    assert torch.isfinite(rw["ref"]).all() and rw["ref"].std() > 1e-3, "the adversarial weights must still give a non-trivial depth map"
    assert cal["l1_total_vs_full_m"] <= cal["tol_total_m"]
    # never worse than the fixed cheap mode -- beyond what the per-site neck calibration is allowed to spend (round 5: it takes weight-only
    # sites until its frame reads AUTO_TOL_NECK_ABS_M against the reference; the fixed mode keeps the whole neck on both products)
    from bodyslam_amd.zoedepth import AUTO_TOL_NECK_CAP_M
    assert l1 <= max(l1w + 2e-5, 1.3 * AUTO_TOL_NECK_CAP_M)          # (the cap both neck stages stop at, round 6)
    # the absolute check made on the device (the chosen modes against the reference-precision engine) must tell the same story as the
    # fp32 oracle on the host: same frame size, other frame
    assert cal["l1_abs_vs_reference_m"] is not None and cal["l1_abs_vs_reference_m"] <= cal["tol_abs_m"] and "warning" not in cal
    assert l1 <= 1e-4, "the north star's tolerance must hold on every adversarial weight set"
    if hook is _hook_outlier_channels:
        # 50x outlier channels that K, V and fc2's input see undamped: the single 16-bit Q / K / V / P of the plain attention kernel cost
        # 2.9e-4 m here (tools/probes/outlier_rounding_study.py) and every cheap GEMM mode more than its tolerance: the guard must turn
        # the split-precision attention on and the GEMM classes back to "full" (round 3: 2.7e-4 m with all of that still single)
        assert cal["attn_mode"] == "corr"
        assert cal["l1_vs_full_m"]["attn:single"] > cal["tol_class_m"]
        assert l1 < 0.5 * l1w


# ------------------------------------------------------------------------------------------------
# Round 6: the calibration judges every stage on several frames and validates its choice on frames it has not
# seen; the evidence that the tolerance holds is taken on frames that NEITHER has seen.
def test_calibration_holds_on_held_out_frames():
    """The bench's weights (random_zoedepth_weights seed 0), calibrated as the bench calibrates them; then 16 consecutive frames of another
    synthetic sequence -- no calibration frame, no hold-out frame -- against the reference-precision engine on the device (itself held to
    1e-5 m of the fp32 oracle: test_reference_precision_engine), and one of them against the oracle directly.  Every frame must meet the
    north star's 1e-4 m; the calibration's own hold-out figures must be inside its line."""
    from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
    from bodyslam_amd.zoedepth import AUTO_CAL_FRAMES, AUTO_HOLDOUT_FRAMES, AUTO_TOL_HOLDOUT_M, TOLERANCE_M, ZoeConfig, ZoeDepthEngine
    from oracle import zoedepth_ref as Z
    cfg = ZoeConfig()
    w = random_zoedepth_weights(cfg, seed=0)
    eng = ZoeDepthEngine(w, cfg, precision="accurate")
    frames = torch.from_numpy(make_sequence(16, 480, 640, seed=77))
    d = eng.infer(frames.cuda())[0].clone()
    cal = eng.calibration
    truth = eng.reference_depth(frames.cuda())
    per = (d - truth).abs().flatten(1).mean(1).cpu()
    report(f"[held-out frames] 16 frames vs the reference-precision engine: mean {per.mean():.3e} max {per.max():.3e} min {per.min():.3e} m; "
           f"calibration: {cal['frames']} frames, worst {cal['l1_abs_vs_reference_m']:.3e}, hold-out {cal.get('holdout')}; "
           f"one-pass share {cal.get('neck_sites', {}).get('flops_share_plain')}, {cal['calibrate_s']} s")
    assert cal["frames"] == AUTO_CAL_FRAMES and cal["holdout"]["frames"] == AUTO_HOLDOUT_FRAMES
    assert cal["holdout"]["l1_max_m"] <= AUTO_TOL_HOLDOUT_M and cal["l1_total_vs_full_m"] <= cal["tol_total_m"]
    assert float(per.max()) <= TOLERANCE_M, f"a held-out frame misses the tolerance: {per.tolist()}"
    assert float(per.max()) <= 1.3 * AUTO_TOL_HOLDOUT_M, "frames the calibration has not seen sit far above the ones it validated on"
    # what the static bias correction of the one-pass products is worth: the same engine, the same frames, the corrections not applied
    if eng.site_bias_corr:
        os.environ["BS_NECK_BIAS_CORR"] = "0"
        try:
            eng._plans.clear()
            d_off = eng.infer(frames[:4].cuda())[0].clone()
        finally:
            del os.environ["BS_NECK_BIAS_CORR"]
            eng._plans.clear()
        per_off = (d_off - truth[:4]).abs().flatten(1).mean(1).cpu()
        report(f"[held-out frames] one-pass products WITHOUT their static bias correction: {per_off.mean():.3e} m (with it {per[:4].mean():.3e}) on frames 0-3; "
               f"{len(eng.site_bias_corr)} products carry one")
        assert float(per[:4].mean()) < float(per_off.mean()), "the static bias correction must not make the one-pass products worse"
    for i in (11,):
        with torch.no_grad():
            ref = Z.infer_depth(w, Z.ZOED_NK, frames[i:i + 1], flip_aug=True)
        l1 = (d[i].cpu() - ref[0]).abs().mean().item()
        report(f"[held-out frames] frame {i} vs the fp32 oracle: {l1:.3e} m (vs the device reference {per[i]:.3e})")
        assert l1 <= TOLERANCE_M and abs(l1 - float(per[i])) <= 1.5e-5


def test_wstat_opt_in_holds_tolerance():
    """BS_AUTO_WSTAT=1: the backbone's rank-1 weight-rounding correction as a static bias row from the calibration frames' channel means (no
    bs_col_mean / bs_rank1_bias launches).  An opt-in (profiles/r06_calibration_experiments.txt (8)): the calibration must pick it where the
    class tolerance allows, the plan must run without the two helper launches for those classes, and unseen frames must hold the tolerance."""
    from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
    from bodyslam_amd.zoedepth import TOLERANCE_M, ZoeConfig, ZoeDepthEngine
    cfg = ZoeConfig()
    w = random_zoedepth_weights(cfg, seed=0)
    os.environ["BS_AUTO_WSTAT"] = "1"
    try:
        eng = ZoeDepthEngine(w, cfg, precision="accurate")
        frames = torch.from_numpy(make_sequence(4, 480, 640, seed=78)).cuda()
        d = eng.infer(frames)[0].clone()
        cal = eng.calibration
        names = eng.plan_for(4, 480, 640, True).plan.names
    finally:
        del os.environ["BS_AUTO_WSTAT"]
    stat = [k for k, v in cal["class_modes"].items() if v == "wstat"]
    assert stat, f"no class took the static form: {cal['class_modes']} ({cal['l1_vs_full_m']})"
    assert len(cal["backbone_bias_corr"]) == len(stat) * cfg.layers and sorted(eng.backbone_bias_corr) == sorted(cal["backbone_bias_corr"])
    for k in stat:
        assert not any(n.endswith(f".{k}.cm") or n.endswith(f".{k}.r1") for n in names), f"class {k} still launches its rank-1 pair"
    truth = eng.reference_depth(frames)
    per = (d - truth).abs().flatten(1).mean(1).cpu()
    report(f"[wstat opt-in] classes {cal['class_modes']}: 4 unseen frames vs the reference-precision engine {per.tolist()}; hold-out {cal['holdout']['l1_max_m']:.3e}")
    assert float(per.max()) <= TOLERANCE_M


def test_calibration_uses_every_frame_it_is_given():
    """calibrate(frames_u8=...) judges on ALL the caller's frames (round 5 silently kept the first) and validates on the caller's hold-out set"""
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    from oracle import zoedepth_ref as Z
    cfg_o = small_oracle_cfg()
    eng = ZoeDepthEngine(Z.synth_weights(cfg_o, seed=4), product_cfg(cfg_o), dtype=torch.float16, target_hw=(96, 128), precision="accurate")
    fr = torch.from_numpy(make_sequence(5, 120, 160, seed=9)).cuda()
    rep = eng.calibrate(frames_u8=fr[:3], holdout_u8=fr[3:])
    assert rep["frames"] == 3 and rep["frame"] == "120x160" and rep["holdout"]["frames"] == 2 and len(rep["holdout"]["l1_frames_m"]) == 2
    assert rep["l1_abs_vs_reference_m"] is not None and rep["l1_abs_vs_reference_m"] <= rep["tol_abs_m"]
    # the static corrections of the one-pass sites travel with the report and are what another engine applies
    eng2 = ZoeDepthEngine(Z.synth_weights(cfg_o, seed=4), product_cfg(cfg_o), dtype=torch.float16, target_hw=(96, 128), precision="accurate")
    eng2.apply_calibration(rep)
    assert eng2.neck_mode == eng.neck_mode and sorted(eng2.site_bias_corr) == sorted(eng.site_bias_corr)
    assert torch.equal(eng.infer(fr[:2])[0], eng2.infer(fr[:2])[0])


@pytest.mark.parametrize("neck", ["full", "one_pass"])
def test_no_launch_reads_memory_the_plan_has_not_written(neck):
    """A plan's intermediates are torch.empty blocks, and producers skip the planes their consumers do not read (lo8 / hi8 planes in front of
    weight-only / one-pass products, the unrouted head's half of the bins tensors).  With the caching allocator's free memory filled with 0xFF
    (NaN in fp16, fp32 and e4m3) before the plan is built, the router logits and the network's depth map must come out bit for bit as from clean
    memory: nothing that reaches the output is read before it is written (tools/probes/poisoned_pool.py runs the same on the full-size network)."""
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine, _ZoePlan
    from oracle import zoedepth_ref as Z
    cfg_o = small_oracle_cfg()
    eng = ZoeDepthEngine(Z.synth_weights(cfg_o, seed=4), product_cfg(cfg_o), dtype=torch.float16, target_hw=(96, 128), precision="accurate",
                         class_modes="full", attn_mode="single", neck_mode="full")
    if neck == "one_pass":
        sites = sorted(k for k in eng.f8s if not (k[0] == "l" and k[1].isdigit()) and k != "pe.w" and not k.endswith("w_cls") and not k.startswith("mh."))
        eng.set_class_modes({}, "wonly:" + ",".join(sites) + ";plain:" + ",".join(k for k in sites if k != "rh.conv2.w"))
    frames = torch.from_numpy(make_sequence(3, 120, 160, seed=9)).cuda()

    def run():
        plan = _ZoePlan(eng, 3, 120, 160, True)
        plan.frames.copy_(frames)
        plan.run(None)
        torch.cuda.synchronize()
        return plan.depth_net.clone(), plan.logits.clone(), plan.depth_m.clone()

    torch.cuda.empty_cache()
    clean = run()
    for pat in (0xFF, 0x3C):
        torch.cuda.empty_cache()
        junk = torch.full((4 << 30,), pat, dtype=torch.uint8, device="cuda")      # what the next plan's buffers are carved from
        del junk
        got = run()
        for a, b, name in zip(clean, got, ("depth_net", "logits", "depth_m")):
            assert torch.isfinite(b).all() and torch.equal(a, b), f"{name} depends on memory the plan did not write (pool pattern 0x{pat:02X}, neck {neck})"


def test_last_launch_reproducible_beside_another_streams_mfma_kernel():
    """Round 6 (profiles/r06_reproducibility.txt (5)-(8)): bs_logbinom_depth_ex in the forms that gathered the embedding's corners with 4- / 8-byte
    LDS reads gave wrong values in the last 16 lanes of a few waves whenever a kernel of ANOTHER stream shared its CUs and issued MFMA -- bs_rank1_bias is the plan's
    one MFMA kernel small enough to do so (4-40 % of the launches beside it, 0 alone; tools/probes/gather_beside_stream.py).  The shipped form reads 16 bytes per
    lane: relaunched on its untouched inputs beside bursts of bs_rank1_bias on a second stream it must give the first launch's bits every time."""
    from bodyslam_amd import _lib as LL
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine, _ZoePlan
    from oracle import zoedepth_ref as Z
    cfg_o = small_oracle_cfg()
    eng = ZoeDepthEngine(Z.synth_weights(cfg_o, seed=4), product_cfg(cfg_o), dtype=torch.float16, target_hw=(96, 128), precision="accurate",
                         class_modes="full", attn_mode="single", neck_mode="full")
    plan = _ZoePlan(eng, 4, 120, 160, True)
    plan.frames.copy_(torch.from_numpy(make_sequence(4, 120, 160, seed=9)).cuda())
    plan.run(None)
    torch.cuda.synchronize()
    base = plan.depth_net.clone()
    P = plan.plan
    fn, args = P.calls[P.names.index("logbinom")]
    lib = LL.load_library()
    g = torch.Generator(device="cpu").manual_seed(0)
    abar = torch.randn(16, 1024, generator=g).to(torch.bfloat16).cuda()
    dw = (torch.randn(3072, 1024, generator=g) * 0.01).to(torch.bfloat16).cuda()
    acc = torch.zeros(16, 3072, device="cuda")
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = n = 0
    for _ in range(20):
        for _ in range(480):
            LL.check(lib.bs_rank1_bias(abar.data_ptr(), dw.data_ptr(), acc.data_ptr(), 16, 3072, 1024, sB.cuda_stream), "bs_rank1_bias")
        outs = []
        with torch.cuda.stream(sA):
            for _ in range(100):
                LL.check(fn(*args, sA.cuda_stream), "logbinom")
                outs.append(plan.depth_net.clone())
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, base)) for o in outs)
        n += len(outs)
    assert bad == 0, f"{bad} of {n} relaunches beside bs_rank1_bias on a second stream differ from the first"
