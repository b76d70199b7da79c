"""Probe helper: a second process that keeps calibrating (engine builds, ~90 four-frame plans built and dropped per calibration: allocation churn + many small
launches) and running a 64-frame plan for N seconds -- the kind of neighbour beside which a forward has been seen to differ from its rerun."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bodyslam_amd.zoedepth as ZD
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cfg = ZD.ZoeConfig()
w = random_zoedepth_weights(cfg, seed=0)
fr = torch.from_numpy(make_sequence(64, 480, 640, seed=1)).cuda()
t0, n = time.time(), 0
open("/tmp/gpu_load_ready", "w").write("1")
while time.time() - t0 < secs:
    ZD._CALIBRATION_CACHE.clear()
    eng = ZD.ZoeDepthEngine(w, cfg, precision="accurate")
    eng.infer(fr)                      # calibrates, then builds and runs the 64-frame plan
    eng.infer(fr)
    torch.cuda.synchronize()
    del eng
    torch.cuda.empty_cache()
    n += 1
print(f"churn: {n} engines calibrated and run in {time.time() - t0:.0f} s")
