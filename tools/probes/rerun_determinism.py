"""Probe: run ONE plan of the two-process test's small network over and over (the same input, the same launches) and compare every marked
intermediate with the first run's: which stage, if any, is not reproducible -- alone and while another process keeps the GPU busy.
    python tools/probes/rerun_determinism.py [iterations] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import two_process_shard as T
from bodyslam_amd.zoedepth import ZoeDepthEngine, _ZoePlan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg_p, wz, wp, frames = T._case()
eng = ZoeDepthEngine(wz, cfg_p, target_hw=T.TARGET, precision="accurate", class_modes="full", attn_mode="single", neck_mode=os.environ.get("NECK", "full"))
plan = _ZoePlan(eng, 4, T.H, T.W, True)
plan.frames.copy_(torch.from_numpy(frames[:4]).cuda())
base = {}
plan.run(base)
torch.cuda.synchronize()
order = list(base)
bad = {}
fresh = os.environ.get("FRESH") == "1"       # a new plan per iteration (new buffers, new descriptors), as the calibration builds them
skip = lambda nm: False
route = None


def view(name, t, meta):
    """the part of a marked tensor the plan has written: the routed head's half of a bins tensor, the hi16 values of a fused map"""
    global route
    if meta and meta[0] == "nhwc_route":
        if route is None:
            route = plan.route.clone().long()
        nb2 = meta[4] // 2
        tt = t.view(meta[1], meta[2], meta[3], 2, nb2)
        return torch.stack([tt[b, :, :, int(route[b])] for b in range(meta[1])])
    if meta and meta[0] == "nhwc" and len(meta) > 5 and meta[5] in (2, 3):
        C = meta[4]
        return t.view(meta[1], meta[2], meta[3], 2 * C)[..., :C]
    return t
for it in range(n):
    if fresh:
        del plan
        plan = _ZoePlan(eng, 4, T.H, T.W, True)
        plan.frames.copy_(torch.from_numpy(frames[:4]).cuda())
    taps = {}
    if fresh and it % 2:
        plan.run(None)                 # the plain path every other time: only the final maps can be compared
        torch.cuda.synchronize()
        taps = {"depth_net": (plan.depth_net.clone(), None)}
    else:
        plan.run(taps)
    torch.cuda.synchronize()
    for name in [n_ for n_ in order if n_ in taps and not skip(n_)]:
        a, b = view(name, base[name][0], base[name][1]), view(name, taps[name][0], taps[name][1] if taps[name][1] is not None else base[name][1])
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            bad.setdefault(name, []).append((it, int((d > 0).sum()), float(d.max())))
            break                      # the first stage that differs in this iteration
print(f"{n} reruns of the tap path: stages that differed first: " + (", ".join(f"{k}: {len(v)}x (e.g. iteration {v[0][0]}: {v[0][1]} elements, max {v[0][2]:.3e})" for k, v in bad.items()) or "none"), flush=True)
# the same through the plain (non-tap) path: only the final map can be compared
ref = None
nbad = 0
for it in range(n):
    plan.run(None)
    torch.cuda.synchronize()
    d = plan.depth_m.clone()
    if ref is None:
        ref = d
    elif not torch.equal(ref, d):
        nbad += 1
print(f"{n} reruns of the plain path: {nbad} final maps differ from the first", flush=True)
