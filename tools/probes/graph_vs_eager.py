"""Probe: the B = 64 depth plan issued launch by launch (484 ctypes calls) vs replayed as one HIP graph: ms per forward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine
cfg = ZoeConfig()
eng = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision="accurate")
zp = eng.plan_for(64, 480, 640, True)
zp.frames.copy_(torch.from_numpy(make_sequence(64, 480, 640, seed=1)).cuda())
def timed(fn, n=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
e = timed(zp.plan.run)
d0 = zp.depth_u16.clone()
zp.plan.capture()
g = timed(zp.plan.run)
print(f"eager {e:.2f} ms, graph replay {g:.2f} ms per forward of 64 frames; same bits: {torch.equal(d0, zp.depth_u16)}")
