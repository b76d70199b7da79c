"""Probe: the reference's whole per-frame loop (BodySlamPipeline.run_slam_loop: depth, MPEM, RGB-D odometry + UKF fusion, chain, pose graph
every 500 frames, TSDF map, back-projection) at the real configuration -- full ZoeD_NK, 640x480, the reference's TSDF parameters -- on
one GPU: frames/s of the whole loop, and of its parts measured separately.
    python tools/probes/slam_loop_full.py [frames=257] [batch=64]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd import _lib as L                                                                      # noqa: E402
from bodyslam_amd.pipeline import BodySlamPipeline                                                      # noqa: E402
from bodyslam_amd.rgbd_odometry import RGBDOdometry                                                     # noqa: E402
from bodyslam_amd.synthetic import make_sequence, random_cyclepose_weights, random_zoedepth_weights    # noqa: E402
from bodyslam_amd.tsdf import TSDF, PinholeCameraIntrinsic, RGBDImage                                   # noqa: E402
from bodyslam_amd.zoedepth import ZoeConfig                                                             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 257
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = ZoeConfig()
pipe = BodySlamPipeline(random_zoedepth_weights(cfg, seed=0), random_cyclepose_weights(seed=0), cfg, batch=B)
frames = torch.from_numpy(make_sequence(N, 480, 640, seed=1)).cuda()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


pipe.run_slam_loop(frames[:B + 2], vo=True, tsdf=TSDF())                          # builds the plans, graphs and a first map (not timed)
torch.cuda.empty_cache()
tsdf = TSDF()
tsdf.reserve(4096)
if os.environ.get("BS_PROFILE"):                                                  # where the host's time goes (python -X importtime is no help here)
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    pipe.run_slam_loop(frames, vo=True, tsdf=TSDF(), posegraph_every=500)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    torch.cuda.empty_cache()
res, t_all = timed(lambda: pipe.run_slam_loop(frames, vo=True, tsdf=tsdf, posegraph_every=500))
tsdf = pipe.last_tsdf
print(f"run_slam_loop, {N} frames 640x480, batch {B}, VO fusion + TSDF map (1 mm voxels, 0.1 m truncation, 32^3 units): {t_all * 1e3:.0f} ms = "
      f"{t_all / N * 1e3:.2f} ms/frame = {N / t_all:.1f} frames/s; map: {tsdf.n_units} units = {tsdf.n_units * 32 ** 3 * 20 / 1e9:.1f} GB of voxels; "
      f"modes {pipe.zoe.class_modes} neck {pipe.zoe.neck_mode!r}")
# the parts, each on its own
(depth, _, t_rel), t_net = timed(lambda: pipe.depth_and_pose_block(frames, 0, N, False, 0))
odo = RGBDOdometry(tuple(pipe.K))
raw = L.depth_u16_to_m(depth.contiguous(), pipe.depth_scale, 3.0e38)
for i in range(3):
    odo.track(frames[i], raw[i])
_, t_vo = timed(lambda: [odo.track(frames[i], raw[i]) for i in range(3, N)])
odo_b = RGBDOdometry(tuple(pipe.K))
odo_b.track_block(frames[:B], raw[:B])
_, t_vob = timed(lambda: [odo_b.track_block(frames[a:a + B], raw[a:a + B]) for a in range(B, N, B)])
t2 = TSDF()
t2.reserve(tsdf.n_units + 4096)
dm = L.depth_u16_to_m(depth.contiguous(), pipe.depth_scale, pipe.depth_trunc)
intr = PinholeCameraIntrinsic(640, 480, *[float(v) for v in pipe.K])
g = res.g_abs.cpu().numpy()


def build():
    for i in range(N):
        t2.build_3D_map(RGBDImage(frames[i], dm[i]), intr, g[i], sync=False)
    return t2.sync()


_, t_map = timed(build)
t3 = TSDF()
t3.reserve(tsdf.n_units + 4096)


def build_batched():
    for a in range(0, N, B):
        t3.build_3D_map_batch([RGBDImage(frames[i], dm[i]) for i in range(a, min(a + B, N))], intr, g[a:a + B])
    return t3.sync()


_, t_mapb = timed(build_batched)
pcd, t_ext = timed(t2.extract_pcd)
mesh, t_mesh = timed(t2.extract_mesh)
print(f"parts: depth + MPEM (batched) {t_net / N * 1e3:.2f} ms/frame; RGB-D odometry track_block() {t_vob / max(N - B, 1) * 1e3:.3f} ms/pair (track(): "
      f"{t_vo / (N - 3) * 1e3:.3f}); TSDF build_3D_map_batch {t_mapb / N * 1e3:.3f} ms/frame (frame by frame, streamed: {t_map / N * 1e3:.3f}); extract_pcd {t_ext * 1e3:.0f} ms -> {pcd.points.shape[0]} points; extract_mesh {t_mesh * 1e3:.0f} ms -> "
      f"{mesh.vertices.shape[0]} vertices, {mesh.triangles.shape[0]} triangles; allocated {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
