"""RGB-D odometry oracle (oracle/rgbd_odometry_ref.py) against rendered ground truth (tests/_render.py): the restated hybrid
Gauss-Newton must recover a known sub-pixel camera motion.  Open3D parity is unpinned (see the oracle's header); this is the pin
the algorithm does have.  CPU only."""
import numpy as np
import pytest

from _render import render, small_pose
from oracle import rgbd_odometry_ref as R

H, W = 120, 160
K = (150.0, 150.0, 80.0, 60.0)


@pytest.mark.parametrize("motion", [(0.004, -0.006, 0.003, 0.002, -0.0015, 0.001), (-0.01, 0.004, -0.008, -0.004, 0.003, -0.002),
                                    (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)])
def test_oracle_recovers_rendered_motion(motion):
    pose_s = small_pose(*motion)
    ct, dt = render(np.eye(4), K, H, W)
    cs, ds = render(pose_s, K, H, W)
    trace = []
    T = R.rgbd_odometry(cs, ds, ct, dt, K, 3.0, trace=trace)
    assert np.abs(T[:3, 3] - pose_s[:3, 3]).max() < 5e-5            # 50 um on motions of 1-4 mm
    assert np.abs(T[:3, :3] - pose_s[:3, :3]).max() < 1e-4
    assert trace[-1][3] <= trace[0][3] * 1.0001 or trace[0][3] < 0.1  # the cost does not grow
    assert len(trace) == 35 and trace[-1][4] > 0.9 * H * W


def test_gauss_newton_step_undoes_a_perturbation():
    """at a pose perturbed from the truth by a small twist d, the first step is ~ -d: Jacobians and signs"""
    pose_s = small_pose(0.004, -0.006, 0.003, 0.002, -0.0015, 0.001)
    ct, dt = render(np.eye(4), K, H, W)
    cs, ds = render(pose_s, K, H, W)
    Is, Ds = R.prepare(cs, ds, 3.0)
    It, Dt = R.prepare(ct, dt, 3.0)
    grads = (*R.sobel(It), *R.sobel(Dt))
    for comp, mag in ((0, 2e-3), (1, 2e-3), (2, 2e-3), (3, 1e-3), (4, 1e-3), (5, 1e-3)):
        d = np.zeros(6)
        d[comp] = mag
        A, b, _, _ = R.accumulate(Is, Ds, It, Dt, grads, K, R.se3_exp(d) @ pose_s)
        step = np.linalg.solve(A, -b) / mag
        expect = np.zeros(6)
        expect[comp] = -1.0
        assert np.abs(step - expect).max() < 0.12, (comp, step)


def test_invalid_depth_and_pyramid():
    _, d = render(np.eye(4), K, H, W)
    d[:20] = 0.0
    I, D = R.prepare(np.zeros((H, W, 3), np.uint8), d, 3.0)
    assert np.isnan(D[:20]).all() and not np.isnan(D[20:]).any()
    D1 = R.pyr_down_depth(D)
    assert D1.shape == (60, 80) and np.isnan(D1[:10]).all() and not np.isnan(D1[10:]).any()
    assert np.allclose(R.pyr_down(np.full((H, W), 0.25)), 0.25)
    gx, gy = R.sobel(np.tile(np.arange(W, dtype=float), (H, 1)) * 0.5)
    assert np.allclose(gx[:, 1:-1], 0.5) and np.allclose(gy, 0.0)


@pytest.mark.parametrize("motion", [(0.01, -0.015, 0.008, 0.012, -0.009, 0.006), (-0.02, 0.01, -0.012, -0.015, 0.01, -0.008)])
def test_nearest_association_recovers_motion_to_a_pixel(motion):
    """association="nearest" + loss="o3d" (Open3D's form, the product's default): the target is read at the rounded pixel, so the
    fixed point sits within about half a pixel of the truth (z / f = 2 mm per pixel here) -- tested on motions of 9-15 mm."""
    pose_s = small_pose(*motion)
    ct, dt = render(np.eye(4), K, H, W)
    cs, ds = render(pose_s, K, H, W)
    trace = []
    T = R.rgbd_odometry(cs, ds, ct, dt, K, 3.0, trace=trace, association="nearest")
    assert np.abs(T[:3, 3] - pose_s[:3, 3]).max() < 1.2e-3
    assert np.abs(T[:3, :3] - pose_s[:3, :3]).max() < 4e-3
    assert len(trace) == 35 and trace[-1][4] > 0.8 * H * W
    assert trace[-1][3] < 0.5 * trace[0][3]                            # the robust cost falls


def test_nearest_first_step_points_at_the_truth():
    """one Gauss-Newton step of the nearest / o3d form from the identity on a half-pixel translation: the linearisation (target gradient
    at the rounded pixel) moves the pose towards the truth although the cost is piecewise constant"""
    pose_s = small_pose(0.0, 0.0, 0.0, 0.001, -0.0008, 0.0)           # 0.5 / 0.4 pixels
    ct, dt = render(np.eye(4), K, H, W)
    cs, ds = render(pose_s, K, H, W)
    Is, Ds = R.prepare(cs, ds, 3.0)
    It, Dt = R.prepare(ct, dt, 3.0)
    grads = (*R.sobel(It), *R.sobel(Dt))
    A, b, _, n = R.accumulate(Is, Ds, It, Dt, grads, K, np.eye(4), association="nearest")
    step = np.linalg.solve(A, -b)
    assert n > 0.95 * H * W
    assert abs(step[3] - 0.001) < 4e-4 and abs(step[4] + 0.0008) < 4e-4 and np.abs(step[:3]).max() < 3e-3
