"""RGB-D odometry on the GPU (bodyslam_amd/rgbd_odometry.py + csrc/odometry.hip) against the numpy oracle step by step, against
rendered ground truth, and through the VO fusion step (reference: BodySLAM_not_refactored/3DM/visual_odometry.py:60-120)."""
import numpy as np
import pytest

from _render import render, small_pose

pytestmark = pytest.mark.gpu

H, W = 120, 160
K = (150.0, 150.0, 80.0, 60.0)


def frames(motion, holes=False):
    pose_s = small_pose(*motion)
    ct, dt = render(np.eye(4), K, H, W)
    cs, ds = render(pose_s, K, H, W)
    if holes:
        rng = np.random.default_rng(3)
        ds[rng.random((H, W)) < 0.05] = 0.0
        dt[40:50, 70:90] = 0.0
    return pose_s, cs, ds, ct, dt


@pytest.mark.parametrize("assoc", ["nearest", "bilinear"])
@pytest.mark.parametrize("holes", [False, True])
def test_matches_oracle_and_truth(holes, assoc):
    """both association modes (nearest + Open3D's robust step: the default; bilinear + IRLS: the option) against the oracle's
    statement of the same mode, step by step, and against the rendered truth"""
    from bodyslam_amd.rgbd_odometry import RGBDOdometry
    from oracle import rgbd_odometry_ref as R
    near = assoc == "nearest"
    pose_s, cs, ds, ct, dt = frames((0.01, -0.015, 0.008, 0.012, -0.009, 0.006) if near else (0.004, -0.006, 0.003, 0.002, -0.0015, 0.001), holes)
    odo = RGBDOdometry(K, association=assoc)
    assert RGBDOdometry(K).association == "nearest"
    # the sums of one step at the same pose.  Not at the identity: there 136 coarse-level pixels project EXACTLY onto the image
    # border and the last bit of fx X / z + cx decides whether they count (1199 of 1200 already in pure fp64) -- a start a hair off
    # the identity has no such ties, so inlier counts must be equal and the sums agree to the precision of the fp32 images
    init = R.se3_exp(np.array([3e-4, -2e-4, 1e-4, 2e-4, 1e-4, -1e-4]))
    odo.estimate(cs, ds, ct, dt, 3.0, init=init, trace=True)
    ref_trace = []
    R.rgbd_odometry(cs, ds, ct, dt, K, 3.0, init=init, trace=ref_trace, association=assoc)
    g, r = odo.last_trace[0], ref_trace[0]
    assert g[0] == r[0] == 2 and g[4] == r[4]
    assert np.abs(g[1] - r[1]).max() <= 2e-5 * np.abs(r[1]).max() and np.abs(g[2] - r[2]).max() <= 2e-5 * np.abs(r[2]).max() + 1e-9
    assert abs(g[3] - r[3]) <= 2e-5 * r[3]
    T = odo.estimate(cs, ds, ct, dt, 3.0, trace=True)
    ref_trace = []
    T_ref = R.rgbd_odometry(cs, ds, ct, dt, K, 3.0, trace=ref_trace, association=assoc)
    assert len(odo.last_trace) == len(ref_trace) == 35
    worst = max(abs(a[4] - b[4]) for a, b in zip(odo.last_trace, ref_trace))
    print(f"{assoc} holes={holes}: largest inlier-count difference over the 35 steps: {worst} pixels; |T - T_oracle| = {np.abs(T - T_ref).max():.2e}")
    # the two runs follow poses that differ at the 1e-7 level, so a few pixels whose bilinear footprint grazes the image border or
    # an invalid-depth hole fall on different sides; bounded here, and immaterial next to the checks on T below
    # (nearest: a projected point within 1e-7 of a half-pixel boundary may round either way: the two runs then read neighbouring
    # pixels for a handful of points, which moves T at the 1e-5 level)
    assert worst <= max(2, int(0.002 * H * W))
    assert np.abs(T - T_ref).max() < (5e-5 if near else 2e-6)
    if near:        # the nearest-pixel fixed point sits within about half a pixel (z / f = 2 mm) of the truth
        assert np.abs(T[:3, 3] - pose_s[:3, 3]).max() < 1.5e-3 and np.abs(T[:3, :3] - pose_s[:3, :3]).max() < 5e-3
    else:
        assert np.abs(T[:3, 3] - pose_s[:3, 3]).max() < (1e-4 if holes else 5e-5)
        assert np.abs(T[:3, :3] - pose_s[:3, :3]).max() < 2e-4
    dev1 = odo.estimate(cs, ds, ct, dt, 3.0)            # the device-side loop (no trace): same steps, solve and pose update in a kernel
    assert np.abs(dev1 - T).max() < 1e-9 and odo.last_trace is None
    assert np.array_equal(odo.estimate(cs, ds, ct, dt, 3.0), dev1)       # fixed-order reduction: run-to-run identical


def test_full_resolution_and_vo_fusion():
    """640x480 with the reference's intrinsics through VO: MPEM's rotation, the filtered odometry translation"""
    from bodyslam_amd.rgbd_odometry import RGBDOdometry
    from bodyslam_amd.tsdf import RGBDImage
    from bodyslam_amd.visual_odometry import VO
    Kf = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)
    pose_s = small_pose(0.002, -0.003, 0.001, 0.0015, -0.001, 0.0008)
    ct, dt = render(np.eye(4), Kf, 480, 640)
    cs, ds = render(pose_s, Kf, 480, 640)
    rel_b = RGBDOdometry(Kf, association="bilinear")(RGBDImage(cs, ds), RGBDImage(ct, dt))
    assert np.abs(np.linalg.inv(rel_b)[:3, 3] - pose_s[:3, 3]).max() < 3e-5
    odo = RGBDOdometry(Kf)                                # the default: nearest-pixel association (half a pixel = 0.4 mm at 0.3 m here)
    rel = odo(RGBDImage(cs, ds), RGBDImage(ct, dt))       # what _compute_vo_o3d returns: the inverse of source -> target
    assert np.abs(np.linalg.inv(rel)[:3, 3] - pose_s[:3, 3]).max() < 6e-4

    class FakeMPEM:
        def infer_relative_pose_between(self, a, b):
            return np.eye(4, dtype=np.float32)

    vo = VO(FakeMPEM(), intrinsic=Kf)                     # no callable given: the built-in odometry is used
    T = vo.estimate_relative_pose_between("a", "b", RGBDImage(ct, dt), RGBDImage(cs, ds), 1)
    gain = 1.1 / 2.1                                       # first UKF update from P0 = 0.1 I, Q = R = I
    assert np.allclose(T[:3, 3], gain * rel[:3, 3], atol=1e-6) and np.allclose(T[:3, :3], np.eye(3))


@pytest.mark.parametrize("assoc", ["nearest", "bilinear"])
def test_track_stream_equals_pairwise_and_is_fast(assoc):
    """track(): consecutive frames, every pyramid built once, the launches of a frame replayed from a HIP graph, no host round trip --
    the same transforms as estimate() pair by pair (bit for bit: the same kernels in the same order), at well under a millisecond
    per 640x480 pair"""
    import time
    import torch
    from bodyslam_amd.rgbd_odometry import RGBDOdometry
    Kf = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)
    poses = [small_pose(0.002 * i, -0.003 * i, 0.001 * i, 0.0015 * i, -0.001 * i, 0.0008 * i) for i in range(5)]
    fr = [render(p, Kf, 480, 640) for p in poses]
    dev = torch.device("cuda:0")
    cols = [torch.from_numpy(c).to(dev) for c, _ in fr]
    deps = [torch.from_numpy(d).to(dev) for _, d in fr]
    odo = RGBDOdometry(Kf, association=assoc)
    got = [odo.track(c, d) for c, d in zip(cols, deps)]
    assert got[0] is None and odo._trk["graph"] is not None
    for i in range(1, 5):
        T = odo.estimate(fr[i][0], fr[i][1], fr[i - 1][0], fr[i - 1][1], 3.0e38)
        assert np.array_equal(got[i].cpu().numpy().reshape(3, 4), T[:3]), i
    # timing: replayed pairs, one synchronisation at the end
    odo.reset()
    odo.track(cols[0], deps[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    out = [odo.track(cols[1 + (i & 1)], deps[1 + (i & 1)]) for i in range(n)]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    import os
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "odometry_track.txt"), "a") as f:
        f.write(f"track() {assoc}: {ms:.3f} ms per 640x480 pair (HIP graph replay, pyramids reused, no readback)\n")
    print(f"track() {assoc}: {ms:.3f} ms per pair")
    assert ms < 1.0


@pytest.mark.parametrize("assoc", ["nearest", "bilinear"])
def test_track_block_equals_pairwise_and_is_fast(assoc):
    """track_block(): the pairs of a block of consecutive frames do not depend on each other (every pair starts from the identity,
    visual_odometry.py:100), so a block is tracked as simultaneous pairs, one launch per stage.  Pair for pair the same transforms as
    estimate() (bit for bit: the same sums in the same order), across block borders too; a 64-frame block of 640x480 frames costs
    well under 0.25 ms per pair"""
    import os
    import time
    import torch
    from bodyslam_amd.rgbd_odometry import RGBDOdometry
    Kf = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)
    poses = [small_pose(0.002 * i, -0.003 * i, 0.001 * i, 0.0015 * i, -0.001 * i, 0.0008 * i) for i in range(6)]
    fr = [render(p, Kf, 480, 640) for p in poses]
    dev = torch.device("cuda:0")
    cols = torch.stack([torch.from_numpy(c) for c, _ in fr]).to(dev)
    deps = torch.stack([torch.from_numpy(d) for _, d in fr]).to(dev)
    odo = RGBDOdometry(Kf, association=assoc)
    a = odo.track_block(cols[:3], deps[:3])                 # frames 0..2: two pairs
    b = odo.track_block(cols[3:4], deps[3:4])               # frame 3 against the kept frame 2
    c = odo.track_block(cols[4:], deps[4:], max_block=1)    # pieces of one frame
    assert a.shape == (2, 12) and b.shape == (1, 12) and c.shape == (2, 12)
    got = torch.cat([a, b, c]).cpu().numpy()
    for i in range(1, 6):
        T = odo.estimate(fr[i][0], fr[i][1], fr[i - 1][0], fr[i - 1][1], 3.0e38)
        assert np.array_equal(got[i - 1].reshape(3, 4), T[:3]), i
    odo.reset()
    assert odo.track_block(cols[:1], deps[:1]).shape == (0, 12) and odo.track_block(cols[:0], deps[:0]).shape == (0, 12)
    # a first call with ONE frame sizes the buffers for one frame; the larger block that follows reallocates them and must still pair its
    # first frame with the one kept (round-3 advisor: the reallocation used to drop it and return n - 1 pairs)
    d = odo.track_block(cols[1:4], deps[1:4])
    assert d.shape == (3, 12) and np.array_equal(d.cpu().numpy(), got[:3])
    odo.reset()
    # timing: blocks of 64 frames
    big_c = cols[torch.arange(64, device=dev) % 6].contiguous()
    big_d = deps[torch.arange(64, device=dev) % 6].contiguous()
    odo.track_block(big_c, big_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = odo.track_block(big_c, big_d)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (3 * 64) * 1e3
    assert np.array_equal(out[1].cpu().numpy(), got[0])     # (pair 1 of the big block = frames 0 -> 1 again)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "odometry_track.txt"), "a") as f:
        f.write(f"track_block() {assoc}: {ms:.3f} ms per 640x480 pair (blocks of 64 frames, one launch per stage, no readback)\n")
    print(f"track_block() {assoc}: {ms:.3f} ms per pair")
    assert ms < 0.25
