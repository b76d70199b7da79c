"""Probe: what the multi-frame calibration chooses, and how the choice holds on frames it has not seen, per weight seed and threshold set.
    python tools/probes/calibration_sweep.py <seed,seed,...> <abs:plain:holdout in 1e-5 m> [<abs:plain:holdout> ...]
For every (seed, thresholds): the chosen modes, the share of the neck's FLOPs on one pass, the calibration's own figures and the per-frame depth
L1 against the reference-precision engine on 8 frames that neither the calibration nor its hold-out check has seen (synthetic seeds 31..38)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bodyslam_amd.zoedepth as ZD  # noqa: E402
from bodyslam_amd.synthetic import WEIGHT_VARIANTS, make_sequence, random_zoedepth_weights  # noqa: E402

seeds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
sets = [tuple(float(v) * 1e-5 for v in a.split(":")) for a in sys.argv[2:]] or [(5.0e-5, 6.0e-5, 7.0e-5)]
variant = os.environ.get("WEIGHTS", "gaussian")
H, W = 480, 640
test_frames = torch.from_numpy(np.concatenate([make_sequence(1, H, W, seed=s) for s in range(31, 39)], 0)).cuda()
cfg = ZD.ZoeConfig()
for seed in seeds:
    w = random_zoedepth_weights(cfg, seed=seed)
    if WEIGHT_VARIANTS[variant] is not None:
        WEIGHT_VARIANTS[variant](w)
    truth = None
    for (t_abs, t_plain, t_hold) in sets:
        ZD.AUTO_TOL_NECK_ABS_M, ZD.AUTO_TOL_NECK_PLAIN_ABS_M, ZD.AUTO_TOL_HOLDOUT_M = t_abs, t_plain, t_hold
        ZD._CALIBRATION_CACHE.clear()
        eng = ZD.ZoeDepthEngine(w, cfg, precision="accurate")
        t0 = time.time()
        cal = eng.calibrate(H, W)
        t_cal = time.time() - t0
        if truth is None:
            truth = eng.reference_depth(test_frames)
        d = eng.infer(test_frames)[0]
        per = (d - truth).abs().flatten(1).mean(1)
        ns = cal.get("neck_sites", {})
        print(f"seed {seed} [{variant}] tol abs {t_abs:.1e} plain {t_plain:.1e} holdout {t_hold:.1e}: {cal['class_modes']} attn {cal['attn_mode']} "
              f"backbone-choice {cal.get('l1_backbone_choice_vs_reference_m') or float('nan'):.2e} floor {cal.get('l1_best_vs_reference_m') or float('nan'):.2e}; "
              f"wonly {ns.get('weight_only')} share {ns.get('flops_share_weight_only')}; plain {ns.get('plain')} share {ns.get('flops_share_plain')}; "
              f"cal worst-frame {cal['l1_abs_vs_reference_m']:.2e}; holdout {cal.get('holdout')}; calibrate {t_cal:.0f} s", flush=True)
        print(f"    trail (L1 worst calibration frame with the candidate added): w-only start {ns.get('l1_weight_only_m')} {ns.get('l1_with_candidate_plain_m')}", flush=True)
        print(f"    alone: {ns.get('l1_alone_vs_chosen_m')}", flush=True)
        print(f"    8 unseen frames vs the reference-precision engine: mean {per.mean().item():.3e} max {per.max().item():.3e} min {per.min().item():.3e}  "
              f"{[round(float(v) * 1e5, 2) for v in per]} (1e-5 m)", flush=True)
        del eng
        torch.cuda.empty_cache()
