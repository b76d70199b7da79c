"""Probe: the TSDF map at the reference's own parameters (1 mm voxels, 0.1 m truncation, 32^3 units, stride 8; 3DM/tsdf.py:6-12)
on 640x480 frames -- units opened per frame, host unit discovery time, integrate kernel time and HBM rate, extraction time."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd.tsdf import TSDF, PinholeCameraIntrinsic, RGBDImage   # noqa: E402

H, W = 480, 640
K = (383.1901395, 383.1901395, 276.4727783203125, 124.3335933685303)          # slam.py:25-28
v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
rng = np.random.default_rng(0)
color = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
intr = PinholeCameraIntrinsic(W, H, *K)
t = TSDF()
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    depth = (0.12 + 0.03 * np.sin(u / 90.0 + 0.3 * f) * np.cos(v / 70.0)).astype(np.float32)      # an endoscopic working distance
    pose = np.eye(4)
    pose[:3, 3] = (0.002 * f, 0.0, 0.001 * f)
    E = np.linalg.inv(pose)
    n0 = len(t.index)
    tw = time.perf_counter()
    t.build_3D_map(RGBDImage(color, depth), intr, E)
    nv = t.last_units * 32 ** 3
    t1 = time.perf_counter()
    print(f"frame {f}: {t.last_units} units touched ({len(t.index) - n0} new), {nv / 1e6:.0f} M voxels = {nv * 20 / 1e9:.2f} GB of voxel state; "
          f"unit discovery (device hash table) {t.last_touch_ms:.3f} ms, integrate kernel {t.last_kernel_ms:.3f} ms (HIP events; upper bound of its "
          f"traffic, 40 B x every voxel of the touched units: {nv * 40 / (t.last_kernel_ms * 1e-3) / 1e12:.2f} TB/s); whole build_3D_map call "
          f"{(t1 - tw) * 1e3:.1f} ms wall", flush=True)
t0 = time.perf_counter()
pcd = t.extract_pcd()
torch.cuda.synchronize()
print(f"extract_pcd: {pcd.points.shape[0]} points from {len(t.index)} units in {(time.perf_counter() - t0) * 1e3:.0f} ms; "
      f"allocated {torch.cuda.memory_allocated() / 1e9:.1f} GB")
