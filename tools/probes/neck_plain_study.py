"""Probe (round 5): which neck / head products tolerate running with NO correction product at all (one 16-bit pass, `f8_skip_from = -1`)
once the per-site calibration has made them weight-only?  Yardstick: the reference-precision engine on the device.  For every large
weight-only site: depth L1 against the reference with that ONE site plain (the others as calibrated), then the accumulation in the order
error per FLOP saved.
    python tools/probes/neck_plain_study.py [seed ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine, _ZoePlan

H, W = 480, 640
cfg = ZoeConfig()
for seed in [int(a) for a in sys.argv[1:]] or (0,):
    wz = random_zoedepth_weights(cfg, seed=seed)
    eng = ZoeDepthEngine(wz, cfg, precision="accurate")
    cal = eng.calibrate(H, W)
    frames = torch.from_numpy(make_sequence(1, H, W, seed=11)).cuda()      # the calibration's own frame
    truth = eng.reference_depth(frames)
    nm = cal["neck_mode"]
    wonly = nm[6:].split(";")[0].split(",") if nm.startswith("wonly:") else []
    print(f"seed {seed}: calibrated {cal['class_modes']} attn {cal['attn_mode']} neck {nm!r}: L1 vs reference {cal['l1_abs_vs_reference_m']:.3e}", flush=True)
    flops = {}

    def depth(plain_sites):
        eng.set_class_modes({}, "wonly:" + ",".join(wonly) + (";plain:" + ",".join(plain_sites) if plain_sites else ""))
        plan = _ZoePlan(eng, 1, H, W, True)
        plan.frames.copy_(frames)
        plan.run(None)
        d = plan.depth_m.clone()
        torch.cuda.synchronize()
        flops.update(plan.site_flops)
        del plan
        return d

    base = depth(())
    print(f"  as calibrated: L1 vs reference {(base - truth).abs().mean().item():.3e}", flush=True)
    tot = sum(flops.values())
    cands = [k for k in wonly if k != "rh.conv2.w"]
    alone = {}
    for k in sorted(cands, key=lambda k_: -flops[k_]):
        d = depth((k,))
        alone[k] = ((d - truth).abs().mean().item(), (d - base).abs().mean().item())
        print(f"  {k:18s} alone plain: L1 vs reference {alone[k][0]:.3e}  vs calibrated {alone[k][1]:.3e}  ({100 * flops[k] / tot:.1f} % of the neck's FLOPs)", flush=True)
    order = sorted(alone, key=lambda k_: alone[k_][1] / flops[k_])
    chosen = []
    for k in order:
        chosen.append(k)
        d = depth(tuple(chosen))
        print(f"  + {k:18s} -> {len(chosen)} sites plain: L1 vs reference {(d - truth).abs().mean().item():.3e}  "
              f"({100 * sum(flops[c] for c in chosen) / tot:.1f} % of the neck's FLOPs on one pass)", flush=True)
    del eng
    torch.cuda.empty_cache()
