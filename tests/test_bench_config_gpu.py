"""The configurations bench.py times, verified at their stated size (BASELINE.json configs 2-4):

  * the B = 64 full-size ZoeD_NK plan (NB = 128 network inputs: 256x256x64 tiles with the FP8 correction stages, the
    side-stream tail split of o_proj / fc2, the two-lane plan under load) against the B = 1 plan -- the plan that
    tests/test_zoedepth_gpu.py compares with the fp32 oracle tap by tap -- on sampled frames, bit for bit;
  * a 256-frame run_sequence on batch 64 against the same sequence cut into world = 2 contiguous blocks, the two ranks
    emulated one after the other on this GPU (relatives stitched on the host in place of the RCCL all-gather);
  * 256 consecutive frame pairs through the CyclePose engine against the oracle on a sample.
Needs an MI355X: `pytest -m gpu`.  (The oracle runs only on single frames / pairs: a 256-frame oracle pass would take an hour.)"""
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "bench_config_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    print(line)


def _free():
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def weights():
    from bodyslam_amd.synthetic import random_cyclepose_weights, random_zoedepth_weights
    from bodyslam_amd.zoedepth import ZoeConfig
    cfg = ZoeConfig()
    return cfg, random_zoedepth_weights(cfg, seed=0), random_cyclepose_weights(seed=0)


@pytest.mark.parametrize("dtype,precision,B", [(torch.float16, "accurate", 128), (torch.float16, "accurate", 64), (torch.float16, "fast", 64),
                                               (torch.bfloat16, "accurate", 64), (torch.bfloat16, "fast", 64)])
def test_bench_batch_equals_single_frame_plan(weights, dtype, precision, B):
    """bench.py's plan (B = 128 since round 5, B = 64 before; 640x480, ZoeD_NK) gives, for sampled frames, exactly the bits of the B = 1 plan."""
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeDepthEngine
    cfg, wz, _ = weights
    eng = ZoeDepthEngine(wz, cfg, dtype=dtype, precision=precision)
    frames = torch.from_numpy(make_sequence(B, 480, 640, seed=0)).cuda()
    dm, du = eng.infer(frames)
    dm, du = dm.clone(), du.clone()
    plan = eng.plan_for(B, 480, 640, True)
    tiles = sorted({gi["tile"] for gi in plan.plan.gemm_info.values()})
    assert 9 in tiles, tiles                                  # the 256x256x64 tile is what the bench's dominant kernel runs
    route64 = plan.route.clone()
    assert torch.isfinite(dm).all() and (dm > 0).all()
    sample = (0, 17, 40, B - 1)
    for i in sample:
        d1, u1 = eng.infer(frames[i:i + 1])
        assert torch.equal(d1[0], dm[i]), f"frame {i}: B={B} plan differs from the B=1 plan (max {(d1[0] - dm[i]).abs().max().item():.3e})"
        assert torch.equal(u1[0], du[i])
        r1 = eng.plan_for(1, 480, 640, True).route
        assert torch.equal(r1, route64[[i, B + i]])           # per-image routing: frame i and its flipped copy
    report(f"B={B} plan == B=1 plan on frames {sample} [{str(dtype)[6:]} {precision}], tiles used {tiles}")
    del eng, plan
    _free()


def test_sharded_sequence_equals_unsharded_256_frames(weights):
    """Config 4's structure at config 2/3's size: 256 frames, batch 64.  world = 1 against world = 2 (blocks [0,128) and
    [128,256), rank 1 reading frame 127 as its halo), every output bit-equal: depth, relatives, absolute poses, point
    counts and indices."""
    from bodyslam_amd.pipeline import BodySlamPipeline, local_pairs, shard_bounds
    from bodyslam_amd.synthetic import make_sequence
    cfg, wz, wp = weights
    N, H, W = 256, 480, 640
    frames = torch.from_numpy(make_sequence(N, H, W, seed=3))
    pipe = BodySlamPipeline(wz, wp, cfg, batch=64, precision="accurate")
    whole = pipe.run_sequence(frames, keep_points=True)
    assert whole.depth_u16.shape == (N, H, W) and whole.t_rel.shape == (N - 1, 4, 4) and whole.g_abs.shape == (N, 4, 4)
    # world = 2: blocks of 128 frames (two full batches each); world = 3: blocks of 86 / 85 / 85 frames, i.e. other batch
    # compositions and ragged last batches that run through the padded 64-frame plan
    for world in (2, 3):
        # rank by rank: stage 1+2 of every rank first (their relatives are what the all-gather would exchange) ...
        blocks = []
        for r in range(world):
            s, e = shard_bounds(N, world, r)
            depth, _, t_loc = pipe.depth_and_pose_block(frames, s, e)
            blocks.append((s, e, depth, t_loc))
            assert t_loc.shape[0] == local_pairs(s, e).shape[0]
        t_all = torch.cat([b[3] for b in blocks], 0)
        assert torch.equal(t_all.view(-1, 4, 4), whole.t_rel), f"world={world}: relatives differ"
        # ... then stage 3 of each rank on the stitched relatives
        for (s, e, depth, _) in blocks:
            res = pipe.chain_and_backproject(N, s, e, depth, None, t_all, keep_points=True)
            assert torch.equal(res.depth_u16, whole.depth_u16[s:e]), f"world={world}: depth of block [{s},{e}) differs"
            assert torch.equal(res.g_abs, whole.g_abs)
            assert torch.equal(res.point_counts, whole.point_counts[s:e])
            for j in (0, 1, (e - s) // 2, e - s - 1):
                xa, ia = res.points[j]
                xb, ib = whole.points[s + j]
                assert torch.equal(ia, ib) and torch.equal(xa, xb)
        if world == 2:
            t_all2 = t_all
    t_all = t_all2
    # run_sequence itself with the gather hook (what a rank executes, with the collective replaced)
    res1 = pipe.run_sequence(frames, rank=1, world=2, gather=lambda t_loc, counts: t_all)
    assert torch.equal(res1.g_abs, whole.g_abs) and torch.equal(res1.depth_u16, whole.depth_u16[128:])
    with pytest.raises(RuntimeError):
        pipe.run_sequence(frames[:8], rank=0, world=2)        # world > 1 without a process group must not chain local poses only
    R = whole.g_abs[:, :3, :3]
    assert (R.transpose(1, 2) @ R - torch.eye(3, dtype=torch.float64, device=R.device)).abs().max().item() < 1e-12
    report(f"256-frame sequence: world=2 and world=3 blocks == unsharded (depth, t_rel, g_abs, counts, indices, points); "
           f"points/frame {whole.point_counts.float().mean().item():.0f}")
    del pipe, whole
    _free()


def test_config4_size_eight_ranks_emulated(weights):
    """BASELINE config 4 at its size: a 1 000-frame sequence cut into the 8 blocks of 125 frames that 8 ranks own (one 64-frame batch +
    a 61-frame ragged one through the padded plan, each rank reading the frame before its block as its halo), run rank by rank on
    one GPU with the all-gather replaced by the stitched relatives: depth, relatives, absolute poses and point counts bit-equal to
    the unsharded run.  (The collective itself is covered by tests/test_sharding_cpu.py, world 2 over gloo; no 8-GPU node was
    available to measure the scaling.)"""
    from bodyslam_amd.pipeline import BodySlamPipeline, local_pairs, shard_bounds
    from bodyslam_amd.synthetic import make_sequence
    cfg, wz, wp = weights
    N, H, W, world = 1000, 480, 640, 8
    frames = torch.from_numpy(make_sequence(N, H, W, seed=6))
    pipe = BodySlamPipeline(wz, wp, cfg, batch=64, precision="accurate")
    whole = pipe.run_sequence(frames)
    assert whole.depth_u16.shape == (N, H, W) and whole.g_abs.shape == (N, 4, 4)
    blocks = []
    for r in range(world):
        s, e = shard_bounds(N, world, r)
        assert e - s == 125
        depth, _, t_loc = pipe.depth_and_pose_block(frames, s, e)
        assert t_loc.shape[0] == local_pairs(s, e).shape[0]
        assert torch.equal(depth, whole.depth_u16[s:e]), f"rank {r}: depth of block [{s},{e}) differs from the unsharded run"
        blocks.append((s, e, t_loc))
        del depth
    t_all = torch.cat([b[2] for b in blocks], 0)
    assert torch.equal(t_all.view(-1, 4, 4), whole.t_rel)
    for r in (0, 3, 7):
        s, e = blocks[r][0], blocks[r][1]
        res = pipe.run_sequence(frames, rank=r, world=world, gather=lambda t_loc, counts: t_all)
        assert torch.equal(res.g_abs, whole.g_abs) and torch.equal(res.depth_u16, whole.depth_u16[s:e])
        assert torch.equal(res.point_counts, whole.point_counts[s:e])
    report(f"1000-frame sequence: 8 emulated ranks of 125 frames == unsharded (depth, t_rel, g_abs, counts)")
    del pipe, whole
    _free()


def test_ragged_block_through_padded_plan(weights):
    """A 70-frame block on batch 64: the 6-frame tail runs through the 64-frame plan (pad_ragged) and equals the result of a
    pipeline that builds a 6-frame plan."""
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    cfg, wz, wp = weights
    frames = torch.from_numpy(make_sequence(70, 480, 640, seed=4))
    pipe = BodySlamPipeline(wz, wp, cfg, batch=64, precision="accurate")
    a = pipe.run_sequence(frames)
    assert (64, 480, 640, True) in pipe.zoe._plans and (6, 480, 640, True) not in pipe.zoe._plans
    pipe.pad_ragged = False
    b = pipe.run_sequence(frames)
    assert (6, 480, 640, True) in pipe.zoe._plans
    assert torch.equal(a.depth_u16, b.depth_u16) and torch.equal(a.t_rel, b.t_rel) and torch.equal(a.g_abs, b.g_abs)
    del pipe
    _free()


@pytest.mark.parametrize("dtype", [torch.float16])
def test_cyclepose_256_pairs(weights, dtype):
    """Config 3: 256 consecutive pairs (257 frames) through CyclePoseEngine in batches of 64; a sample of pairs against the
    oracle, all poses rigid, and the batched result equal to pair-at-a-time calls."""
    from bodyslam_amd.cyclepose import CyclePoseEngine
    from bodyslam_amd.synthetic import make_sequence
    from oracle import cyclepose_ref as CP
    _, _, wp = weights
    frames = torch.from_numpy(make_sequence(257, 480, 640, seed=6))
    eng = CyclePoseEngine(wp, dtype=dtype)
    Ts = []
    for b0 in range(0, 256, 64):
        chunk = frames[b0: b0 + 65].cuda()
        pairs = torch.tensor([[i, i + 1] for i in range(64)], dtype=torch.int32, device="cuda")
        Ts.append(eng.infer_pairs(chunk, pairs).clone())
    T = torch.cat(Ts).cpu()
    assert T.shape == (256, 4, 4)
    sample = [0, 63, 64, 129, 255]
    x = CP.center_crop_pair(frames, torch.tensor([[i, i + 1] for i in sample]))
    with torch.no_grad():
        Tref = CP.forward_pose(wp, x)
    err = (T[sample] - Tref).abs().max().item()
    R = T[:, :3, :3]
    orth = (R @ R.transpose(1, 2) - torch.eye(3)).abs().max().item()
    one = eng.infer_pairs(frames[129:131].cuda(), torch.tensor([[0, 1]], dtype=torch.int32, device="cuda")).cpu()[0]
    report(f"cyclepose 256 pairs {dtype}: max|T - T_oracle| on {sample} = {err:.3e}, |R R^T - I| = {orth:.1e}, "
           f"batched vs single pair {(one - T[129]).abs().max().item():.1e}")
    assert err < POSE_TOL
    assert orth < 1e-5
    assert torch.equal(one, T[129])


def test_config5_size_on_one_gpu(weights):
    """BASELINE config 5 at its size, as far as one GPU goes: the BATCHED 1280x1024 plan (B = 16; a 416x512 network input, 833 tokens:
    image blocks straddle GEMM tiles, the per-row bias2 path) against the B = 1 plan that tests/test_zoedepth_gpu.py checks against
    the oracle, bit for bit; and stage 3 of a rank at the configuration's length -- the fp64 chain over 4 000 poses with the
    pose-graph step every 500 (3DM/slam.py:54,159-165: odometry edges only, so the chain comes back untouched), then the
    back-projection of the rank's own 1280x1024 frames with their poses out of the 4 000."""
    from bodyslam_amd import geom3d
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    from oracle import geom3d_ref as G
    cfg, wz, wp = weights
    H, W, B = 1024, 1280, 16
    frames = torch.from_numpy(make_sequence(B, H, W, seed=8))
    pipe = BodySlamPipeline(wz, wp, cfg, batch=B, precision="accurate")
    dm, du = pipe.zoe.infer(frames.cuda())
    dm, du = dm.clone(), du.clone()
    assert torch.isfinite(dm).all() and (dm > 0).all()
    for i in (0, 7, 15):
        d1, u1 = pipe.zoe.infer(frames[i:i + 1].cuda())
        assert torch.equal(d1[0], dm[i]) and torch.equal(u1[0], du[i]), f"frame {i}: B=16 1280x1024 plan differs from the B=1 plan"
    # stage 3 at 4 000 poses: this rank owns frames [3984, 4000)
    N = 4000
    rng = np.random.default_rng(5)
    t_rel = np.tile(np.eye(4, dtype=np.float32), (N - 1, 1, 1))
    ang = rng.normal(0, 2e-3, (N - 1, 3)).astype(np.float32)
    t_rel[:, 0, 1], t_rel[:, 1, 0], t_rel[:, 0, 2], t_rel[:, 2, 0], t_rel[:, 1, 2], t_rel[:, 2, 1] = -ang[:, 2], ang[:, 2], ang[:, 1], -ang[:, 1], -ang[:, 0], ang[:, 0]
    t_rel[:, :3, 3] = rng.normal(0, 1e-3, (N - 1, 3)).astype(np.float32)
    t_all = torch.from_numpy(t_rel).cuda().view(-1, 16)
    pipe.posegraph_every = 500
    res = pipe.chain_and_backproject(N, N - B, N, du, None, t_all, keep_points=True)
    ref = G.pose_chain(t_rel)
    assert np.abs(res.g_abs.cpu().numpy() - ref).max() < 1e-9               # the pose graph left the chain untouched
    dun = du.cpu().numpy().view(np.uint16)
    for j in (0, B - 1):
        rx, ri = G.backproject(dun[j], pose=ref[N - B + j])
        xyz, idx = res.points[j]
        assert np.array_equal(idx.cpu().numpy(), ri) and np.allclose(xyz.cpu().numpy(), rx, atol=1e-5)
    report(f"config 5 on one GPU: B=16 1280x1024 plan == B=1 plan; 4000-pose chain + pose graph every 500 == oracle chain; "
           f"points/frame {res.point_counts.float().mean().item():.0f}")
    del pipe
    _free()


def test_config5_sequence_on_one_gpu(weights):
    """BASELINE config 5 as a SEQUENCE on one GPU: 64 consecutive 1280x1024 frames through run_sequence (MDEM B = 16 per step, MPEM on the
    reference's CenterCrop(128) of a 1280x1024 frame -- top 448, left 576, mpem_interface.py:40-44 --, chain, pose-graph step every 500,
    back-projection): MPEM against the oracle at that size, the chain against the oracle chain, depth of sampled frames bit-equal to
    the B = 1 plan, points against the C oracle.  (The 8-GPU sharding of this configuration is config 4's mechanism at another frame
    size: tests/test_sharding_cpu.py, test_config4_size_eight_ranks_emulated.)"""
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    from oracle import cyclepose_ref as CP
    from oracle import geom3d_ref as G
    cfg, wz, wp = weights
    H, W, B, N = 1024, 1280, 16, 64
    frames = torch.from_numpy(make_sequence(N, H, W, seed=12))
    pipe = BodySlamPipeline(wz, wp, cfg, batch=B, precision="accurate")
    pipe.posegraph_every = 500
    res = pipe.run_sequence(frames, keep_points=True, keep_depth_m=True)
    assert res.t_rel.shape == (N - 1, 4, 4) and res.g_abs.shape == (N, 4, 4) and res.depth_u16.shape == (N, H, W)
    # MPEM at 1280x1024 against the oracle (pairs across batch borders included)
    sample = [0, 14, 15, 16, 40, 62]
    with torch.no_grad():
        Tref = CP.forward_pose(wp, CP.center_crop_pair(frames, torch.tensor([[i, i + 1] for i in sample])))
    perr = (res.t_rel.cpu()[sample] - Tref).abs().max().item()
    # the chain (float64, SO(3) projection per step) against the oracle's chain of the SAME relatives; the pose-graph step leaves it alone
    ref_g = G.pose_chain(res.t_rel.cpu().numpy())
    gerr = np.abs(res.g_abs.cpu().numpy() - ref_g).max()
    # depth: the batched plan against the B = 1 plan (which tests/test_zoedepth_gpu.py::test_other_frame_geometries holds to the oracle)
    for i in (0, 17, 63):
        d1, u1 = pipe.zoe.infer(frames[i:i + 1].cuda())
        assert torch.equal(d1[0], res.depth_m[i]) and torch.equal(u1[0], res.depth_u16[i]), f"frame {i}: sequence depth differs from the B=1 plan"
    du = res.depth_u16.cpu().numpy().view(np.uint16)
    for j in (0, 31, 63):
        rx, ri = G.backproject(du[j], pose=ref_g[j])
        xyz, idx = res.points[j]
        assert np.array_equal(idx.cpu().numpy(), ri) and np.allclose(xyz.cpu().numpy(), rx, atol=1e-5)
    report(f"config 5 sequence on one GPU: 64 frames 1280x1024, MPEM max|T - T_oracle| = {perr:.3e}, chain max err {gerr:.1e}, "
           f"points/frame {res.point_counts.float().mean().item():.0f}")
    assert perr < POSE_TOL and gerr < 1e-9
    del pipe
    _free()


POSE_TOL = 1e-5      # the accurate MPEM path (split-precision convolutions; DESIGN.md, Numerics)
