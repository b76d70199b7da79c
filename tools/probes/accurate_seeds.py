"""Probe: accurate-mode depth L1 vs the oracle over several weight / frame seeds (full-size ZoeD_NK, 640x480)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_zoedepth_gpu as T
from oracle import zoedepth_ref as Z
for seed in [int(a) for a in sys.argv[1:]] or (2, 3, 4):
    r = T.run_case(Z.ZOED_NK, torch.float16, B=1, H=480, W=640, target_hw=(384, 512), seed=seed, precision="accurate")
    e = (r["dm"] - r["ref"]).abs()
    print(f"seed {seed}: L1 {e.mean().item():.3e} max {e.max().item():.3e} signed {(r['dm'] - r['ref']).mean().item():+.3e} route {r['route_p'].tolist()} modes {(r['calibration'] or {}).get('class_modes')} attn {(r['calibration'] or {}).get('attn_mode')} neck {(r['calibration'] or {}).get('neck_mode')!r} abs-vs-device-reference {(r['calibration'] or {}).get('l1_abs_vs_reference_m') or float('nan'):.2e} "
          f"(vs all-full {(r['calibration'] or {}).get('l1_total_vs_full_m', float('nan')):.2e})", flush=True)
