"""Probe: does running the depth network as TWO concurrent half-batches (two plans on two streams: the tail of one kernel's last round is
filled by the other stream's kernels) beat ONE full batch?  Full ZoeD_NK, 640x480, accurate mode, B = 64 (128 forwards with flip-aug).
    python tools/probes/two_stream_step.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights                              # noqa: E402
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine                                             # noqa: E402

cfg = ZoeConfig()
wz = random_zoedepth_weights(cfg, seed=0)
frames = torch.from_numpy(make_sequence(64, 480, 640, seed=1)).cuda()
REPS = 4


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3


one = ZoeDepthEngine(wz, cfg)
p64 = one.plan_for(64, 480, 640, True)
p64.frames.copy_(frames)
t_one = timed(p64.plan.run)
d64 = p64.depth_u16.clone()
modes = dict(one.class_modes)
del one, p64
torch.cuda.empty_cache()
halves = [ZoeDepthEngine(wz, cfg, class_modes=modes) for _ in range(2)]
plans = [h.plan_for(32, 480, 640, True) for h in halves]
for i, p in enumerate(plans):
    p.frames.copy_(frames[32 * i: 32 * i + 32])
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def seq():
    for p in plans:
        p.plan.run()


def par():
    cur = torch.cuda.current_stream()
    for s, p in zip(streams, plans):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            p.plan.run()
    for s in streams:
        cur.wait_stream(s)


t_seq = timed(seq)
t_par = timed(par)
same = torch.equal(torch.cat([p.depth_u16 for p in plans]), d64)
print(f"depth network, 64 frames (128 forwards): one B=64 plan {t_one:.1f} ms; two B=32 plans one after the other {t_seq:.1f} ms; "
      f"two B=32 plans on two streams {t_par:.1f} ms; same depth as the B=64 plan: {same}")
