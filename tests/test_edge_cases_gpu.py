"""Edge cases of the hot path: empty and minimal inputs, all-invalid / all-valid depth maps, a one-frame sequence (no pose pair),
ragged last batches.  Needs an MI355X: `pytest -m gpu`."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def small_cfgs():
    from bodyslam_amd.zoedepth import ZoeConfig
    from oracle import zoedepth_ref as Z
    cfg_o = Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    return cfg_o, ZoeConfig(**{k: v for k, v in dataclasses.asdict(cfg_o).items() if k in names})


def test_backproject_all_invalid_and_all_valid():
    from bodyslam_amd import geom3d
    z = torch.zeros(2, 48, 64, dtype=torch.int16, device="cuda")
    z[1] = 1500
    xyz, idx, cnt = geom3d.backproject(z)
    assert cnt.tolist() == [0, 48 * 64]
    assert torch.equal(idx[1], torch.arange(48 * 64, dtype=torch.int32, device="cuda"))
    # values at and above the truncation (3.0 m * 1000) and the uint16 maximum are invalid
    z2 = torch.tensor([[[2999, 3000, -1, 1]]], dtype=torch.int16, device="cuda")          # -1 = 65535 as uint16
    _, idx2, cnt2 = geom3d.backproject(z2)
    assert cnt2.tolist() == [2] and idx2[0, :2].tolist() == [0, 3]


def test_pose_chain_empty_and_single():
    from bodyslam_amd import geom3d
    g = geom3d.pose_chain(np.zeros((0, 4, 4), np.float32)).cpu().numpy()
    assert g.shape == (1, 4, 4) and np.array_equal(g[0], np.eye(4))
    t = np.eye(4, dtype=np.float32)[None].copy()
    t[0, :3, 3] = [0.1, -0.2, 0.3]
    g = geom3d.pose_chain(t).cpu().numpy()
    assert g.shape == (2, 4, 4) and np.allclose(g[1, :3, 3], [0.1, -0.2, 0.3], atol=1e-7)


def test_sequence_of_one_frame_and_ragged_batches():
    """N = 1: no pair, the chain is the identity; N = 5 with batch 2: a ragged last batch; both against N = 5 in one batch."""
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    from oracle import cyclepose_ref as CP
    from oracle import zoedepth_ref as Z
    cfg_o, cfg_p = small_cfgs()
    wz, wp = Z.synth_weights(cfg_o, seed=4), CP.synth_weights(seed=4)
    frames = make_sequence(5, 160, 192, seed=6)
    one = BodySlamPipeline(wz, wp, cfg_p, batch=2, target_hw=(64, 96)).run_sequence(frames[:1], keep_points=True)
    assert one.t_rel.shape[0] == 0 and one.g_abs.shape == (1, 4, 4) and torch.equal(one.g_abs[0].cpu(), torch.eye(4, dtype=torch.float64))
    assert one.depth_u16.shape == (1, 160, 192) and len(one.points) == 1
    a = BodySlamPipeline(wz, wp, cfg_p, batch=2, target_hw=(64, 96)).run_sequence(frames, keep_depth_m=True)
    b = BodySlamPipeline(wz, wp, cfg_p, batch=5, target_hw=(64, 96)).run_sequence(frames, keep_depth_m=True)
    assert torch.equal(a.depth_u16, b.depth_u16) and torch.equal(a.depth_m, b.depth_m)
    assert torch.equal(a.t_rel, b.t_rel) and torch.equal(a.g_abs, b.g_abs) and torch.equal(a.point_counts, b.point_counts)
    assert torch.equal(one.depth_u16[0], a.depth_u16[0])


def test_empty_batch_calls_are_noops():
    """B = 0 / rows = 0 through the C ABI: success, nothing launched."""
    from bodyslam_amd import _lib as L
    L.init(0)
    x = torch.zeros(4, 64, device="cuda")
    o = torch.zeros(4, 128, device="cuda", dtype=torch.float16)
    lib = L.load_library()
    assert lib.bs_cast_split(L.p(x), L.p(o), 0, 64, L.dt(o), L.stream_ptr()) == 0
    assert lib.bs_relu_split(L.p(o), L.p(o), 0, 64, L.dt(o), L.stream_ptr()) == 0
    g, b = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    assert lib.bs_layernorm(L.p(x), L.p(g), L.p(b), L.p(o), None, 0, 64, 1e-6, L.dt(o), L.stream_ptr()) == 0
    z = torch.zeros(1, 8, 8, dtype=torch.int16, device="cuda")
    from bodyslam_amd import geom3d
    xyz, idx, cnt = geom3d.backproject(z[:0])
    assert xyz.shape[0] == 0 and cnt.numel() == 0
    # the engines: an empty frame batch / an empty pair list return empty results
    from bodyslam_amd.cyclepose import CyclePoseEngine
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    from oracle import cyclepose_ref as CP
    from oracle import zoedepth_ref as Z
    cfg_o, cfg_p = small_cfgs()
    eng = ZoeDepthEngine(Z.synth_weights(cfg_o, seed=1), cfg_p, target_hw=(64, 96), precision="fast")
    dm, du = eng.infer(torch.zeros(0, 160, 192, 3, dtype=torch.uint8, device="cuda"))
    assert dm.shape == (0, 160, 192) and du.shape == (0, 160, 192)
    pose = CyclePoseEngine(CP.synth_weights(seed=1))
    T = pose.infer_pairs(torch.zeros(2, 160, 192, 3, dtype=torch.uint8, device="cuda"), torch.zeros(0, 2, dtype=torch.int32, device="cuda"))
    assert T.shape == (0, 4, 4)


@pytest.mark.parametrize("H,W", [(123, 157), (97, 211), (200, 120)])
def test_odd_frame_sizes_match_oracle(H, W):
    """frame sizes that are not multiples of anything (odd reflect pads, non-square resize ratios, portrait): depth vs the oracle"""
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    from oracle import zoedepth_ref as Z
    cfg_o, cfg_p = small_cfgs()
    w = Z.synth_weights(cfg_o, seed=8)
    frames = torch.from_numpy(make_sequence(2, H, W, seed=8))
    eng = ZoeDepthEngine(w, cfg_p, target_hw=(96, 128), precision="accurate")
    dm, du = eng.infer(frames.cuda())
    with torch.no_grad():
        ref = Z.infer_depth(w, cfg_o, frames, out_hw=(96, 128))
    l1 = (dm.cpu() - ref).abs().mean().item()
    lsb = np.abs(du.cpu().numpy().view(np.uint16).astype(np.int32) - Z.to_uint16(ref).astype(np.int32))
    assert dm.shape == (2, H, W) and l1 < 1e-4 and lsb.max() <= 1, (l1, lsb.max())
