"""VO fusion (bodyslam_amd/visual_odometry.py; reference BodySLAM_not_refactored/3DM/visual_odometry.py:14-93): the restated UKF
against the linear Kalman filter it must equal for identity models (oracle/ukf_ref.py), and the fusion step's call pattern with
stand-in MPEM / odometry objects.  CPU only: nothing here touches the GPU."""
import numpy as np
import pytest

from bodyslam_amd.visual_odometry import VO, MerweScaledSigmaPoints, UnscentedKalmanFilter, unscented_transform
from oracle.ukf_ref import LinearKF


def test_sigma_points_and_weights():
    sp = MerweScaledSigmaPoints(n=3, alpha=1.0, beta=2.0, kappa=3)            # visual_odometry.py:27
    assert sp.lam == 3.0 and np.isclose(sp.Wm.sum(), 1.0)
    assert np.isclose(sp.Wm[0], 0.5) and np.isclose(sp.Wc[0], 2.5) and np.allclose(sp.Wm[1:], 1 / 12)
    rng = np.random.default_rng(0)
    A = rng.normal(size=(3, 3))
    P, x = A @ A.T + np.eye(3), rng.normal(size=3)
    s = sp.sigma_points(x, P)
    m, C = unscented_transform(s, sp.Wm, sp.Wc)
    assert np.allclose(m, x, atol=1e-12) and np.allclose(C, P, atol=1e-12)   # the transform of the identity is exact


def test_ukf_equals_linear_kalman_filter():
    sp = MerweScaledSigmaPoints(n=3, alpha=1.0, beta=2.0, kappa=3)
    ukf = UnscentedKalmanFilter(3, 3, 1, lambda x, dt=None: x, lambda x: x, sp)
    ukf.P *= 0.1
    kf = LinearKF()
    rng = np.random.default_rng(1)
    for k in range(60):
        z = np.array([0.01 * k, -0.02, 0.005 * np.sin(k)]) + 0.001 * rng.normal(size=3)
        ukf.predict(rng.normal(size=3))            # whatever goes in as dt is ignored by the identity model
        kf.predict()
        ukf.update(z)
        kf.update(z)
        assert np.allclose(ukf.x, kf.x, atol=1e-12) and np.allclose(ukf.P, kf.P, atol=1e-12)
    # steady state: with Q = R = I the gain settles at the golden-ratio value (1 + sqrt 5) / (3 + sqrt 5)
    assert np.allclose(np.diag(ukf.K), (1 + 5 ** 0.5) / (3 + 5 ** 0.5), atol=1e-9)


class FakeMPEM:
    def __init__(self):
        self.calls = []

    def infer_relative_pose_between(self, a, b):
        self.calls.append((a, b))
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float32)
        T[:3, 3] = (9.0, 9.0, 9.0)                  # a translation the fusion must NOT use
        return T


def test_fusion_step_uses_mpem_rotation_and_filtered_odometry_translation():
    def odo(curr, prev):
        T = np.eye(4)
        T[:3, 3] = (0.01, 0.0, 0.02)
        return T
    mp = FakeMPEM()
    vo = VO(mp, rgbd_odometry=odo)
    kf = LinearKF()
    for i in range(1, 6):
        T = vo.estimate_relative_pose_between(f"f{i - 1}", f"f{i}", "rgbd_prev", "rgbd_curr", i)
        kf.predict()
        kf.update([0.01, 0.0, 0.02])
        assert np.allclose(T[:3, :3], [[0, -1, 0], [1, 0, 0], [0, 0, 1]])
        assert np.allclose(T[:3, 3], kf.x, atol=1e-6) and not np.allclose(T[:3, 3], 9.0)
    assert mp.calls[0] == ("f0", "f1") and len(mp.calls) == 5


def test_missing_odometry_fails_loudly():
    vo = VO(FakeMPEM())
    with pytest.raises(NotImplementedError, match="RGB-D odometry"):
        vo.estimate_relative_pose_between("a", "b", None, None, 1)
    with pytest.raises(NotImplementedError):
        vo.estimate_relative_pose_between("a", "b", None, None, 1, rgbd_odo=False)
