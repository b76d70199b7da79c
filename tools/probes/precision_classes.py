"""Probe: which backbone GEMM classes need the split-precision corrections?  Full-size ZoeD_NK, 640x480, accurate mode with
the correction products of one or more classes ("qkv", "o", "fc1", "fc2") reduced ("w": weight correction only, "a": activation
correction only, "single": none); depth L1 against the fp32 oracle.
    python tools/probes/precision_classes.py [seeds...]        (on the GPU box; ~1 min per configuration)"""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_zoedepth_gpu as T                      # noqa: E402
from bodyslam_amd.zoedepth import ZoeDepthEngine   # noqa: E402
from oracle import zoedepth_ref as Z               # noqa: E402

ALL = ("qkv", "o", "fc1", "fc2")
WC = {c: "wcls" for c in ALL}
WM = {c: "wmean" for c in ALL}                       # one 16-bit pass on the patch tiles + the rank-1 weight-rounding correction
WM3 = dict(WM, fc1="wcls")                           # ... except fc1, whose weight-rounding error is not token-independent (GELU)
CONFIGS = [(WC, "full"), (WM3, "full"), (WM, "full"), (dict(WC, qkv="wmean", o="wmean"), "full"), (dict(WC, fc2="wmean"), "full")]
seeds = [int(a) for a in sys.argv[1:]] or [1, 2]
out = open(os.path.join(ROOT, "gpurun_out", "precision_classes.txt"), "a")
for seed in seeds:
    w, frames, taps_o, logits, ref, t_or = T.oracle_case(Z.ZOED_NK, 1, 480, 640, (384, 512), seed, 0.0, True)
    for single, neck in CONFIGS:
        eng = ZoeDepthEngine(w, T.product_cfg(Z.ZOED_NK), dtype=torch.float16, precision="accurate", class_modes=single, neck_mode=neck)
        dm, _ = eng.infer(frames.cuda())
        e = (dm.cpu() - ref).abs()
        line = f"seed {seed} class_modes={single} neck={neck}: L1 {e.mean().item():.3e} max {e.max().item():.3e} signed {(dm.cpu() - ref).mean().item():+.3e}"
        print(line, flush=True)
        out.write(line + "\n")
        out.flush()
        del eng
        gc.collect()
        torch.cuda.empty_cache()
