"""Probe: does any launch of a ZoeDepth plan read a buffer the plan has not written?  A fresh plan's intermediates are torch.empty blocks; here the caching
allocator's free memory is filled with a byte pattern before every plan is built (0xFF: NaN in fp16 / fp32 / e4m3; then 0x3C, 0x01), the plan is run through its tap
path and every marked intermediate is compared with a clean run's: the first stage that differs names the reader.
    python tools/probes/poisoned_pool.py [small|full] [accurate|fast]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine, _ZoePlan
which = sys.argv[1] if len(sys.argv) > 1 else "small"
prec = sys.argv[2] if len(sys.argv) > 2 else "accurate"
if which == "small":
    import two_process_shard as T
    cfg, wz, _, frames = T._case()
    H, W, tgt, B = T.H, T.W, T.TARGET, 4
    frames = torch.from_numpy(frames[:B]).cuda()
else:
    from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
    cfg = ZoeConfig()
    wz = random_zoedepth_weights(cfg, seed=0)
    H, W, tgt, B = 480, 640, (384, 512), 2
    frames = torch.from_numpy(make_sequence(B, H, W, seed=3)).cuda()
modes = [dict(class_modes="full", attn_mode="single", neck_mode="full")]
if prec == "accurate":
    modes += [dict(class_modes="wmean", attn_mode="single", neck_mode="w"), dict(class_modes="wcls", attn_mode="corr", neck_mode="full")]
for kw in modes:
    eng = ZoeDepthEngine(wz, cfg, target_hw=tgt, precision=prec, **(kw if prec == "accurate" else {}))
    if prec == "accurate" and kw["neck_mode"] == "w":
        # every neck product on one pass, the form the calibration's second stage produces (producers skip the planes nobody reads)
        sites = sorted(k for k in eng.f8s if not (k[0] == "l" and k[1].isdigit()) and k != "pe.w" and not k.endswith("w_cls") and not k.startswith("mh."))
        eng.set_class_modes({}, "wonly:" + ",".join(sites) + ";plain:" + ",".join(k for k in sites if k != "rh.conv2.w"))
    torch.cuda.empty_cache()
    plan = _ZoePlan(eng, B, H, W, True)
    plan.frames.copy_(frames)
    base = {}
    plan.run(base)
    torch.cuda.synchronize()
    del plan
    for pat in (0xFF, 0x3C, 0x01):
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info()
        junk = torch.full((min(int(free * 0.5), 40 << 30),), pat, dtype=torch.uint8, device="cuda")     # fills what the next allocations will be carved from
        del junk
        plan = _ZoePlan(eng, B, H, W, True)
        plan.frames.copy_(frames)
        taps = {}
        plan.run(taps)
        torch.cuda.synchronize()
        first = None
        # (marks that legitimately hold unwritten bytes: the unrouted head's half of the bins tensors, the planes a one-pass consumer does not read)
        final_only = [n_ for n_ in base if n_ in ("depth_net", "logits", "embed") or n_.startswith("layer")]
        for name in final_only:
            a, b = base[name][0], taps[name][0]
            if not torch.equal(a, b) and not (torch.isnan(a.float()) == torch.isnan(b.float())).all() or not torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())):
                d = (torch.nan_to_num(a.float()) - torch.nan_to_num(b.float())).abs()
                first = (name, int((d > 0).sum()) + int((torch.isnan(a.float()) != torch.isnan(b.float())).sum()), float(d.max()))
                break
        print(f"[{which} {prec} {kw if prec == 'accurate' else ''}] pool poisoned with 0x{pat:02X}: " + ("the backbone stages, the router logits and the network's depth map equal the clean run" if first is None else
              f"FIRST DIFFERING STAGE {first[0]}: {first[1]} elements, max |diff| {first[2]:.3e}"), flush=True)
        del plan, taps
    del eng
