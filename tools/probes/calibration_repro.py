"""Probe: is the calibration of the two-process test's small network reproducible from process to process?  Each run: a fresh process builds the
engine of tests/two_process_shard.py, calibrates, prints the floor (best mode vs the reference-precision engine), the modes and a checksum of a depth map."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import two_process_shard as T
from bodyslam_amd.zoedepth import ZoeDepthEngine
cfg_p, wz, wp, frames = T._case()
eng = ZoeDepthEngine(wz, cfg_p, target_hw=T.TARGET, precision="accurate")
cal = eng.calibrate(T.H, T.W)
d = eng.infer(torch.from_numpy(frames[:2]).cuda())[0]
h = hashlib.blake2b(d.cpu().numpy().tobytes(), digest_size=6).hexdigest()
ns = cal.get("neck_sites", {})
print(f"floor {cal.get('l1_best_vs_reference_m'):.4e} backbone {cal.get('l1_backbone_choice_vs_reference_m'):.4e} abs {cal['l1_abs_vs_reference_m']:.4e} "
      f"classes {'/'.join(cal['class_modes'].values())} attn {cal['attn_mode']} wonly {len(ns.get('weight_only', []))} plain {len(ns.get('plain', []))} depth {h} "
      f"l1_vs_full {({k: round(v, 7) for k, v in cal['l1_vs_full_m'].items()})}", flush=True)
