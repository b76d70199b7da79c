"""Probe: BodySlamPipeline.run_slam_loop runs the network on the caller's stream and the loop's small kernels (RGB-D odometry, back-projection, chain,
TSDF) on a second stream BESIDE the next batch's network -- kernels of two streams share CUs.  After profiles/r06_reproducibility.txt (8) (a kernel's
four-byte LDS gathers disturbed by another stream's MFMA kernel): are the loop's results the same bits run after run?
    python tools/probes/slam_loop_determinism.py [runs] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bodyslam_amd.zoedepth as ZD
from bodyslam_amd.pipeline import BodySlamPipeline
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights, random_cyclepose_weights
from bodyslam_amd.tsdf import TSDF
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 192
cfg = ZD.ZoeConfig()
pipe = BodySlamPipeline(random_zoedepth_weights(cfg, seed=0), random_cyclepose_weights(seed=0), cfg, batch=64)
frames = make_sequence(N, 480, 640, seed=1)
first, diff = None, {}
for r in range(runs):
    t = TSDF(device=0)
    t.reserve(4096)
    res = pipe.run_slam_loop(frames, vo=True, tsdf=t, posegraph_every=500)
    torch.cuda.synchronize()
    w = 0.0
    cur = dict(depth_u16=res.depth_u16.clone(), t_rel=res.t_rel.clone(), g_abs=res.g_abs.clone(), point_counts=res.point_counts.clone(),
               map=torch.tensor([int(pipe.last_tsdf.n_units), int(pipe.last_tsdf.frames_integrated)]))
    if first is None:
        first = cur
    else:
        for k in cur:
            if not torch.equal(cur[k].cpu(), first[k].cpu()):
                d = (cur[k].double().cpu() - first[k].double().cpu()).abs()
                diff.setdefault(k, []).append((r, int((d > 0).sum()), float(d.max())))
    del t, res
    pipe.last_tsdf = None
    torch.cuda.empty_cache()
print(f"{runs} runs of run_slam_loop over {N} frames (batch 64, VO fusion, TSDF at the reference's parameters): " +
      ("every result equals the first run's bit for bit" if not diff else "; ".join(f"{k}: {len(v)} runs differ (e.g. run {v[0][0]}: {v[0][1]} elements, max {v[0][2]:.3e})" for k, v in diff.items())), flush=True)
