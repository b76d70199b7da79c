"""CPU oracle for the VO fusion filter -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file.  The reference (BodySLAM_not_refactored/3DM/visual_odometry.py:27-36,78-90) runs a filterpy
UnscentedKalmanFilter with identity process and measurement models, Q = R = I (filterpy's defaults), P0 = 0.1 I.  filterpy is not
vendored, not in the reference's requirements.txt and not installed here: **parity unpinned**.  For identity models the unscented
transform is exact (sum Wm = 1 reproduces the mean, sum Wc (s - x)(s - x)^T = P reproduces the covariance for any symmetric sigma
set of a linear map), so the filter must equal the linear Kalman filter below -- an independent statement of the same numbers:

    predict:  x <- x,  P <- P + Q          update:  S = P + R,  K = P S^-1,  x <- x + K (z - x),  P <- P - K S K^T
"""
import numpy as np


class LinearKF:
    def __init__(self, n=3, p0=0.1):
        self.x, self.P, self.Q, self.R = np.zeros(n), np.eye(n) * p0, np.eye(n), np.eye(n)

    def predict(self, _ignored=None):
        self.P = self.P + self.Q

    def update(self, z):
        S = self.P + self.R
        K = self.P @ np.linalg.inv(S)
        self.x = self.x + K @ (np.asarray(z, dtype=np.float64) - self.x)
        self.P = self.P - K @ S @ K.T
