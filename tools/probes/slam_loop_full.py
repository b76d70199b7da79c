"""Probe: the reference's whole per-frame loop (run_slam_loop: depth, MPEM, VO fusion, chain, back-projection, TSDF map) at the real
configuration -- full ZoeD_NK, 640x480, the reference's TSDF parameters -- on one GPU, with the time of each stage.
    python tools/probes/slam_loop_full.py [frames=33] [batch=16]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd.pipeline import BodySlamPipeline                                                      # noqa: E402
from bodyslam_amd.synthetic import make_sequence, random_cyclepose_weights, random_zoedepth_weights    # noqa: E402
from bodyslam_amd.tsdf import TSDF                                                                      # noqa: E402
from bodyslam_amd.zoedepth import ZoeConfig                                                             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 33
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = ZoeConfig()
pipe = BodySlamPipeline(random_zoedepth_weights(cfg, seed=0), random_cyclepose_weights(seed=0), cfg, batch=B)
frames = torch.from_numpy(make_sequence(N, 480, 640, seed=1)).cuda()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


pipe.run_sequence(frames[:B + 1])                                             # builds the plans (not timed)
(depth, depth_m, t_rel), t_net = timed(lambda: pipe.depth_and_pose_block(frames, 0, N, False, 0))
t_fused, t_vo = timed(lambda: pipe.fuse_vo(frames, depth, t_rel))
res, t_chain = timed(lambda: pipe.chain_and_backproject(N, 0, N, depth, depth_m, t_fused, False, None))
tsdf = TSDF()
_, t_map = timed(lambda: pipe.integrate_tsdf(tsdf, frames, res))
pcd, t_ext = timed(tsdf.extract_pcd)
du = depth.view(torch.int16).cpu().numpy().view("uint16")
print(f"{N} frames 640x480, batch {B}: depth + MPEM (batched) {t_net * 1e3:.0f} ms = {t_net / N * 1e3:.2f} ms/frame; VO fusion {t_vo * 1e3:.0f} ms = "
      f"{t_vo / (N - 1) * 1e3:.2f} ms/pair; chain + back-projection {t_chain * 1e3:.1f} ms; TSDF map {t_map * 1e3:.0f} ms = {t_map / N * 1e3:.2f} ms/frame "
      f"({tsdf.n_units} units = {tsdf.n_units * 32 ** 3 * 20 / 1e9:.1f} GB of voxels); extract_pcd {t_ext * 1e3:.0f} ms -> {pcd.points.shape[0]} points")
print(f"whole loop: {(t_net + t_vo + t_chain + t_map) / N * 1e3:.2f} ms/frame = {N / (t_net + t_vo + t_chain + t_map):.1f} frames/s; depth PNG range {du.min()}..{du.max()} "
      f"(x 1/1000 = {du.min() / 1000:.3f}..{du.max() / 1000:.3f} m as 3DM reads it); |t_rel - t_fused| translations max {float((t_rel.view(-1, 4, 4)[:, :3, 3] - t_fused.view(-1, 4, 4)[:, :3, 3]).abs().max()):.4f}; "
      f"allocated {torch.cuda.memory_allocated() / 1e9:.1f} GB")
