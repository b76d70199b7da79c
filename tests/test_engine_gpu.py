"""Engine files (bs_zoedepth_forward / bs_cyclepose_forward for a host without Python): a plan exported by
bodyslam_amd/engine_export.py and run through csrc/engine.hip must give, bit for bit, what the Python-driven plan gives."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def small_cfg():
    from bodyslam_amd.zoedepth import ZoeConfig
    from oracle import zoedepth_ref as Z
    cfg_o = Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    return cfg_o, ZoeConfig(**{k: v for k, v in dataclasses.asdict(cfg_o).items() if k in names})


@pytest.mark.parametrize("precision", ["accurate", "fast"])
def test_zoedepth_engine_equals_the_python_plan(tmp_path, precision):
    from bodyslam_amd.engine import Engine
    from bodyslam_amd.engine_export import export_zoedepth
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    from oracle import zoedepth_ref as Z
    cfg_o, cfg_p = small_cfg()
    eng = ZoeDepthEngine(Z.synth_weights(cfg_o, seed=3), cfg_p, precision=precision)
    B, H, W = 2, 480, 640
    path = str(tmp_path / "zoe.bseng")
    info = export_zoedepth(eng, B, H, W, path)                 # (before the plan has ever run: its static buffers are still clean)
    assert info["ops"] > 50 and info["constant_bytes"] > 0 and os.path.getsize(path) < info["constant_bytes"] + (4 << 20)
    frames = torch.from_numpy(make_sequence(2 * B, H, W, seed=9)).cuda()
    e = Engine(path)
    assert e.io("frames")[1] == B * H * W * 3 and e.io("depth_m")[1] == B * H * W * 4 and e.io("depth_u16")[1] == B * H * W * 2
    for k in range(2):                                         # two batches through the same static buffers
        fr = frames[k * B:(k + 1) * B]
        dm, du = eng.infer(fr)
        dm, du = dm.clone(), du.clone()
        em, eu = e.zoedepth_forward(fr)
        assert torch.isfinite(em).all() and (em > 0).all()
        assert torch.equal(em, dm) and torch.equal(eu, du), f"batch {k}: the engine differs from the plan it was exported from"
    with pytest.raises(Exception, match="built for"):
        e.zoedepth_forward(frames[:1])                         # another batch size: refused, not overrun
    e.close()
    # a truncated file is refused
    bad = str(tmp_path / "bad.bseng")
    with open(path, "rb") as f, open(bad, "wb") as g:
        g.write(f.read(4096))
    with pytest.raises(Exception, match="truncated|engine"):
        Engine(bad)


def test_cyclepose_engine_equals_the_python_plan(tmp_path):
    from bodyslam_amd.cyclepose import CyclePoseEngine
    from bodyslam_amd.engine import Engine
    from bodyslam_amd.engine_export import export_cyclepose
    from bodyslam_amd.synthetic import make_sequence
    from oracle import cyclepose_ref as CP
    eng = CyclePoseEngine(CP.synth_weights(seed=4), precision="accurate")
    N, P, H, W = 5, 4, 480, 640
    path = str(tmp_path / "pose.bseng")
    export_cyclepose(eng, N, P, H, W, path)
    frames = torch.from_numpy(make_sequence(N, H, W, seed=11)).cuda()
    pairs = torch.tensor([[i, i + 1] for i in range(P)], dtype=torch.int32, device="cuda")
    T_ref = eng.infer_pairs(frames, pairs).clone()
    e = Engine(path)
    T = e.cyclepose_forward(frames, pairs)
    assert torch.equal(T, T_ref.view(-1, 4, 4))
    assert np.allclose(T[:, 3].cpu().numpy(), [[0, 0, 0, 1]] * P)
    with pytest.raises(Exception, match="no output|no input"):
        e.zoedepth_forward(frames)                             # the wrong model's call: refused (a pose engine has no depth output)
