"""Probe helper: keep the GPU busy from a second process for N seconds (a B = 32 fast-mode ZoeD_NK plan in a loop): the load beside which
tools/probes/rerun_determinism.py and calibration_repro.py are run.   python tools/probes/gpu_load.py [seconds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cfg = ZoeConfig()
eng = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision="fast")
fr = torch.from_numpy(make_sequence(32, 480, 640, seed=1)).cuda()
eng.infer(fr)
torch.cuda.synchronize()
open("/tmp/gpu_load_ready", "w").write("1")
t0, n = time.time(), 0
while time.time() - t0 < secs:
    eng.infer(fr)
    n += 1
    if n % 4 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print(f"load: {n} forwards of 32 frames in {time.time() - t0:.0f} s")
