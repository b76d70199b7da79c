#!/bin/bash
# The headline's spread over weight sets: bench.py on eight seeds of the random-init ZoeDepth weights -- frames/s, the modes the
# calibration chose, its figures on the calibration and hold-out frames.   bash tools/probes/seed_throughput.sh > profiles/rNN_seed_throughput.txt
cd "$(dirname "$0")/../.."
for s in ${SEEDS:-0 1 2 3 4 5 6 7 8}; do
    python3 bench.py --single-mode --no-pmc-traffic --no-slam-loop --no-cpu-baseline --weights-seed $s --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
c = d['calibration'] or {}
ns = c.get('neck_sites', {})
h = c.get('holdout', {})
print(f\"seed {d['weights_seed']}: {d['value']:7.2f} frames/s  classes {'/'.join(c['class_modes'].values())} attn {c['attn_mode']}  \"
      f\"weight-only {len(ns.get('weight_only', []))} sites ({ns.get('flops_share_weight_only')}), one pass {len(ns.get('plain', []))} ({ns.get('flops_share_plain')})  \"
      f\"L1 vs the device reference: backbone choice {c.get('l1_backbone_choice_vs_reference_m', float('nan')):.2e}, calibration frames (worst of {c.get('frames')}) {c['l1_abs_vs_reference_m']:.2e}, \"
      f\"hold-out (worst of {h.get('frames')}) {h.get('l1_max_m', float('nan')):.2e}  conv stack {d['roofline_conv_stack']['frac']:.3f}  calibrate {c.get('calibrate_s')} s\", flush=True)
"
done
