"""Probe: the calibration report of a weight set, in full.   WEIGHTS=outlier python tools/probes/calibration_report.py [seed]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bodyslam_amd.zoedepth as ZD
from bodyslam_amd.synthetic import WEIGHT_VARIANTS, random_zoedepth_weights
variant = os.environ.get("WEIGHTS", "gaussian")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = ZD.ZoeConfig()
w = random_zoedepth_weights(cfg, seed=seed)
if WEIGHT_VARIANTS[variant] is not None:
    WEIGHT_VARIANTS[variant](w)
eng = ZD.ZoeDepthEngine(w, cfg, precision="accurate")
t0 = time.time()
cal = eng.calibrate(480, 640)
cal = {k: (v if k not in ("site_bias_corr", "backbone_bias_corr") else sorted(v)) for k, v in cal.items()}
print(f"[{variant} seed {seed}] calibrate {time.time() - t0:.1f} s")
print(json.dumps(cal, indent=1))
