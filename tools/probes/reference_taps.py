"""Probe: per-tap errors of a ZoeDepthEngine mode against the fp32 oracle on (optionally hooked) weights.
    python tools/probes/reference_taps.py [reference|accurate] [outlier|none] [attn_mode]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_zoedepth_gpu as T                      # noqa: E402
from oracle import zoedepth_ref as Z               # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "reference"
hook = T._hook_outlier_channels if (len(sys.argv) < 3 or sys.argv[2] == "outlier") else None
kw = {}
if prec == "accurate":
    kw = dict(class_modes="full", neck_mode="full", attn_mode=sys.argv[3] if len(sys.argv) > 3 else "corr")
r = T.run_case(Z.ZOED_NK, torch.float16, B=1, H=480, W=640, target_hw=(384, 512), seed=9, precision=prec, weights_hook=hook, **kw)
e = r["dm"] - r["ref"]
print(f"[{prec} {kw} hook {getattr(hook, '__name__', None)}] depth L1 {e.abs().mean():.3e} max {e.abs().max():.3e} signed {e.mean():+.3e}")
T.compare_taps(r["taps_p"], r["taps_o"], None, f"{prec}")
