"""Probe (round 5, VERDICT r4 #1c): which INDIVIDUAL neck / head products tolerate the weight-rounding correction only ("w": 1.5
pass-equivalents instead of 2)?  Device-side only: the yardstick is the reference-precision engine (three 16-bit passes everywhere)
built from the same weights.  For every FP8-format neck weight: depth L1 against the reference with that ONE site weight-only and
everything else on both products; then the greedy accumulation by milliseconds saved.
    python tools/probes/neck_site_study.py [seed ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine, _ZoePlan

# isolated launch times at B = 64 (profiles/r04_plan_call_times.txt), ms: what a site's second correction product is worth (a quarter of it)
SITE_MS = {"rh.projection.w": 10.27 * 4 / 3, "rh.conv1.w": 5.59 * 4 / 3, "fu3.r1.c2.w": 3.61, "fu3.r2.c2.w": 3.27, "nc0.w": 3.25, "fu3.r1.c1.w": 3.07,
           "fu3.r2.c1.w": 3.06, "nc1.w": 1.53, "pj3.c1.w": 1.36, "ra0.up.w": 0.89, "fu2.r1.c2.w": 0.89, "ra3.down.w": 0.85, "fu2.r2.c2.w": 0.83,
           "nc2.w": 0.81, "fu2.r1.c1.w": 0.78, "fu2.r2.c1.w": 0.76, "fu3.proj.w": 0.71, "ra1.up.w": 0.61, "ro0.w_tok": 0.48, "ro1.w_tok": 0.5,
           "ro2.w_tok": 0.48, "ro3.w_tok": 0.48, "ra3.proj.w": 0.41, "ra2.proj.w": 0.40, "pj2.c1.w": 0.38, "rh.conv2.w": 3.43 * 4 / 3}
H, W = 480, 640
cfg = ZoeConfig()
for seed in [int(a) for a in sys.argv[1:]] or (0,):
    wz = random_zoedepth_weights(cfg, seed=seed)
    eng = ZoeDepthEngine(wz, cfg, precision="accurate", class_modes="wmean", attn_mode="single", neck_mode="full")
    frames = torch.from_numpy(make_sequence(1, H, W, seed=11 + seed)).cuda()
    truth = eng.reference_depth(frames)
    sites = [k for k in eng.f8s if not (k[0] == "l" and k[1].isdigit()) and k != "pe.w" and not k.endswith("w_cls")]

    def depth(wonly_sites):
        keep = ",".join(k for k in sites if k not in wonly_sites) or "none"
        eng.set_class_modes({}, keep if wonly_sites else "full")
        plan = _ZoePlan(eng, 1, H, W, True)
        plan.frames.copy_(frames)
        plan.run(None)
        d = plan.depth_m.clone()
        torch.cuda.synchronize()
        del plan
        return d

    base = depth(())
    l1_base = (base - truth).abs().mean().item()
    print(f"seed {seed}: everything on both products: L1 vs reference {l1_base:.3e} m", flush=True)
    alone = {}
    for k in sorted(sites, key=lambda k_: -SITE_MS.get(k_, 0.0)):
        if SITE_MS.get(k, 0.0) < 0.3:
            continue
        d = depth((k,))
        alone[k] = ((d - truth).abs().mean().item(), (d - base).abs().mean().item())
        print(f"  {k:18s} alone weight-only: L1 vs reference {alone[k][0]:.3e}  vs all-full {alone[k][1]:.3e}  saves ~{SITE_MS[k] / 4:.2f} ms", flush=True)
    # greedy: cheapest error per millisecond first
    order = sorted(alone, key=lambda k_: alone[k_][1] / (SITE_MS[k_] / 4))
    chosen, saved = [], 0.0
    for k in order:
        d = depth(tuple(chosen + [k]))
        l1 = (d - truth).abs().mean().item()
        saved_k = SITE_MS[k] / 4
        print(f"  + {k:18s} -> {len(chosen) + 1} sites weight-only: L1 vs reference {l1:.3e}  (saved so far {saved + saved_k:.2f} ms)", flush=True)
        chosen.append(k)
        saved += saved_k
    del eng
    torch.cuda.empty_cache()
