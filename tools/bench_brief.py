#!/usr/bin/env python3
"""Run bench.py (arguments passed through) as a child and print the few numbers an A/B needs."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", *sys.argv[1:]], capture_output=True, text=True)
lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
if not lines:
    print(out.stdout[-2000:], out.stderr[-4000:])
    sys.exit(1)
b = json.loads(lines[-1])
r = b["roofline"]
print(f"{b['precision']}: {b['value']} frames/s, {b['ms_per_step']} ms/step, L1 {b.get('depth_l1_vs_oracle_m')}, hbm {b.get('hbm_allocated_gb')} GB")
print(f"  dominant {r['kernel']}: frac {r['frac']} executed {r.get('executed_frac')} avg {r.get('avg_launch_us')} us traffic {r.get('traffic')}")
print(f"  conv stack: {b.get('roofline_conv_stack')}")
o = b.get("other_mode")
if o:
    print(f"{o['precision']}: {o['value']} frames/s, {o['ms_per_step']} ms/step, L1 {o.get('depth_l1_vs_oracle_m')}, frac {o['roofline']['frac']}")
for k, v in sorted(b.get("kernels", {}).items(), key=lambda kv: -kv[1].get("share_of_step", 0))[:int(__import__("os").environ.get("TOP", "12"))]:
    print(f"  {k:36s} n={v['launches']:4d} avg {v['avg_us']:9.1f} us share {v.get('share_of_step', 0):.3f} exec {v.get('executed_tflops', 0):7.1f} TF")
