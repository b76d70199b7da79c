"""Probe: time of one RGB-D odometry pair at 640x480 (35 Gauss-Newton steps on a 3-level pyramid), and the accuracy on rendered motion."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _render import render, small_pose                      # noqa: E402
from bodyslam_amd.rgbd_odometry import RGBDOdometry        # noqa: E402

K = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)
pose = small_pose(0.002, -0.003, 0.001, 0.0015, -0.001, 0.0008)
ct, dt = render(np.eye(4), K, 480, 640)
cs, ds = render(pose, K, 480, 640)
odo = RGBDOdometry(K)
odo.estimate(cs, ds, ct, dt, 3.0)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    T = odo.estimate(cs, ds, ct, dt, 3.0)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"640x480 pair: {min(ts) * 1e3:.1f} ms best of 5 ({np.mean(ts) * 1e3:.1f} mean), 35 steps; translation error {np.abs(T[:3, 3] - pose[:3, 3]).max() * 1e6:.1f} um "
      f"on a {np.linalg.norm(pose[:3, 3]) * 1e3:.2f} mm motion, rotation error {np.abs(T[:3, :3] - pose[:3, :3]).max():.1e}")
