"""VO fusion of the SLAM loop: drop-in for BodySLAM_not_refactored/3DM/visual_odometry.py:14-93.

The reference's ``VO.estimate_relative_pose_between`` (called at 3DM/slam.py:144, between MPEM and the pose chain) takes the 4x4
from MPEM, takes a translation from Open3D's multi-scale RGB-D odometry (``_compute_vo_o3d`` :97-120: Hybrid method, 20 / 10 / 5
iterations, inverse of the estimated transform), runs both through a 3-state unscented Kalman filter (filterpy, Merwe sigma points
alpha = 1, beta = 2, kappa = 3, P0 = 0.1 I, identity process and measurement models, :27-36) and overwrites the translation of
MPEM's matrix with the filter state (:90).  Built here:

  * the filter -- ``MerweScaledSigmaPoints`` and ``UnscentedKalmanFilter`` restated from their published form (Wan & van der Merwe
    2000; the predict / update sequence of filterpy 1.4: sigma points are re-drawn from the predicted mean and covariance before
    the update) with filterpy's defaults Q = I, R = I.  filterpy is neither vendored with the reference nor listed in its
    requirements.txt, and is not installed here: parity against it is unpinned; tests/test_vo_cpu.py checks the filter against
    the linear Kalman filter it must equal for identity models (oracle/ukf_ref.py).
  * the fusion step with the reference's exact call pattern, including its quirk: ``self.ukf.predict(transformation[:3, 3])`` hands
    MPEM's translation to filterpy as the ``dt`` argument, which the identity process model ignores -- so MPEM contributes the
    rotation only and the translation is the filtered RGB-D odometry displacement.  Kept as is.

  * the RGB-D odometry: bodyslam_amd/rgbd_odometry.py (``RGBDOdometry``: dense hybrid photometric + geometric Gauss-Newton on a
    3-level pyramid with 20 / 10 / 5 iterations, the scheme of Open3D's Hybrid method; HIP kernels in csrc/odometry.hip).  It is an
    implementation of the published scheme, not a restatement of Open3D's source: parity with Open3D is unpinned and looser than
    for the filter; what pins it is rendered ground truth (tests/).  ``VO`` uses it by default when ``intrinsic`` = (fx, fy, cx, cy)
    is given; ``rgbd_odometry=callable(curr_rgbd, prev_rgbd) -> 4x4`` overrides it (e.g. with Open3D's own, where installed).
    Without either, ``estimate_relative_pose_between`` raises -- it never substitutes anything silently.

NOT built: the sparse-feature scaling path (``rgbd_odo=False``: scaling_system.compute_scaling_factor, OpenCV).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


class MerweScaledSigmaPoints:
    """2n + 1 scaled sigma points and their weights (filterpy.kalman.MerweScaledSigmaPoints)"""

    def __init__(self, n: int, alpha: float, beta: float, kappa: float):
        self.n, self.alpha, self.beta, self.kappa = n, alpha, beta, kappa
        lam = alpha ** 2 * (n + kappa) - n
        c = 0.5 / (n + lam)
        self.Wc = np.full(2 * n + 1, c)
        self.Wm = np.full(2 * n + 1, c)
        self.Wc[0] = lam / (n + lam) + (1.0 - alpha ** 2 + beta)
        self.Wm[0] = lam / (n + lam)
        self.lam = lam

    def num_sigmas(self) -> int:
        return 2 * self.n + 1

    def sigma_points(self, x: np.ndarray, P: np.ndarray) -> np.ndarray:
        n = self.n
        U = np.linalg.cholesky((self.lam + n) * P).T          # upper factor, U^T U = (lam + n) P (scipy.linalg.cholesky's default)
        s = np.zeros((2 * n + 1, n))
        s[0] = x
        for k in range(n):
            s[k + 1] = x + U[k]
            s[n + k + 1] = x - U[k]
        return s


def unscented_transform(sigmas: np.ndarray, Wm: np.ndarray, Wc: np.ndarray, noise_cov: Optional[np.ndarray] = None):
    x = Wm @ sigmas
    y = sigmas - x[None, :]
    P = y.T @ (Wc[:, None] * y)
    if noise_cov is not None:
        P = P + noise_cov
    return x, P


class UnscentedKalmanFilter:
    """filterpy.kalman.UnscentedKalmanFilter, the subset the reference uses: fx(x, dt), hx(x), predict(dt), update(z)"""

    def __init__(self, dim_x: int, dim_z: int, dt, fx: Callable, hx: Callable, points: MerweScaledSigmaPoints):
        self.x = np.zeros(dim_x)
        self.P = np.eye(dim_x)
        self.Q = np.eye(dim_x)
        self.R = np.eye(dim_z)
        self._dim_x, self._dim_z, self._dt = dim_x, dim_z, dt
        self.fx, self.hx, self.points_fn = fx, hx, points
        self.Wm, self.Wc = points.Wm, points.Wc
        self.sigmas_f = np.zeros((points.num_sigmas(), dim_x))
        self.sigmas_h = np.zeros((points.num_sigmas(), dim_z))
        self.K = np.zeros((dim_x, dim_z))
        self.y = np.zeros(dim_z)

    def predict(self, dt=None) -> None:
        if dt is None:
            dt = self._dt
        sigmas = self.points_fn.sigma_points(self.x, self.P)
        for i, s in enumerate(sigmas):
            self.sigmas_f[i] = self.fx(s, dt)
        self.x, self.P = unscented_transform(self.sigmas_f, self.Wm, self.Wc, self.Q)
        self.sigmas_f = self.points_fn.sigma_points(self.x, self.P)      # re-drawn to reflect the predicted covariance
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()

    def update(self, z) -> None:
        z = np.asarray(z, dtype=np.float64)
        for i, s in enumerate(self.sigmas_f):
            self.sigmas_h[i] = self.hx(s)
        zp, S = unscented_transform(self.sigmas_h, self.Wm, self.Wc, self.R)
        Pxz = (self.sigmas_f - self.x[None, :]).T @ (self.Wc[:, None] * (self.sigmas_h - zp[None, :]))
        self.K = Pxz @ np.linalg.inv(S)
        self.y = z - zp
        self.x = self.x + self.K @ self.y
        self.P = self.P - self.K @ S @ self.K.T
        self.S = S


class VO:
    def __init__(self, path_to_model, intrinsic_t=None, intrinsic=None, rgbd_odometry: Optional[Callable] = None, **mpem_kwargs):
        """path_to_model: a CyclePose checkpoint path, or any object with ``infer_relative_pose_between(prev, curr) -> 4x4`` (an
        MPEMInterface).  rgbd_odometry(curr_rgbd, prev_rgbd) -> 4x4: the RGB-D odometry (see the module header)."""
        if hasattr(path_to_model, "infer_relative_pose_between"):
            self.mpem_interface = path_to_model
        else:
            from .mpem import MPEMInterface
            self.mpem_interface = MPEMInterface(path_to_model, **mpem_kwargs)
        self.intrinsic_t, self.intrinsic = intrinsic_t, intrinsic
        self.rgbd_odometry = rgbd_odometry
        self.baseline = np.eye(4)
        self.scale_factor = np.array([0, 0, 0])
        state_dim, measurement_dim = 3, 3
        sigma_points = MerweScaledSigmaPoints(n=state_dim, alpha=1.0, beta=2.0, kappa=3)
        self.ukf = UnscentedKalmanFilter(dim_x=state_dim, dim_z=measurement_dim, dt=1, fx=self.state_transition_function,
                                         hx=self.measurement_function, points=sigma_points)
        self.ukf.x = np.zeros(state_dim)
        self.ukf.P *= 0.1

    def state_transition_function(self, translation_vector, dt=None):
        return translation_vector

    def measurement_function(self, rgbd_translation):
        return rgbd_translation

    def compute_scale_factor(self, t_vector, t_scale_vector):
        scale_factor, residuals, rank, s = np.linalg.lstsq(np.diag(t_vector), t_scale_vector, rcond=None)
        return scale_factor

    def estimate_relative_pose_between(self, prev_frame, curr_frame, prev_rgbd, curr_rgbd, i, rgbd_odo: bool = True) -> np.ndarray:
        transformation = np.array(self.mpem_interface.infer_relative_pose_between(prev_frame, curr_frame))
        if not rgbd_odo:
            raise NotImplementedError("the sparse-feature scaling path (scaling_system.compute_scaling_factor, OpenCV) is not built")
        disp = self._compute_vo_o3d(curr_rgbd, prev_rgbd)[:3, 3]
        self.ukf.predict(transformation[:3, 3])        # (sic: the translation goes in as `dt`, which the identity model ignores)
        self.ukf.update(disp)
        transformation[:3, 3] = self.ukf.x
        return transformation

    def _compute_vo_o3d(self, curr_rgbd, ref_rgbd) -> np.ndarray:
        if self.rgbd_odometry is None:
            if self.intrinsic is None or len(tuple(self.intrinsic)) != 4:
                raise NotImplementedError("RGB-D odometry needs VO(..., intrinsic=(fx, fy, cx, cy)) for the built-in RGBDOdometry, or "
                                          "VO(..., rgbd_odometry=callable(curr_rgbd, prev_rgbd) -> 4x4)")
            from .rgbd_odometry import RGBDOdometry
            self.rgbd_odometry = RGBDOdometry(tuple(self.intrinsic))
        return np.asarray(self.rgbd_odometry(curr_rgbd, ref_rgbd), dtype=np.float64)
