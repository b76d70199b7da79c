"""Probe: every KIND of launch of the small plan (layer 0 / level 0 of each) as the victim, relaunched on its untouched inputs beside the one trigger profiles/r06_reproducibility.txt (8)
found -- bs_rank1_bias launches of a full-size plan on a second stream -- and every tensor it is handed compared byte for byte with the first launch's.  Launches that are not idempotent
(they update their output in place: residual GEMMs, the rank-1 update) are recognised by relaunching them alone and skipped.
    BODYSLAM_HIP_LIB=.../libbodyslam_hip_diag.so BS_LOGBINOM_INTERLEAVED=5 python tools/probes/victim_scan.py [relaunches per launch]
(with the diagnostics build's four-byte-gather form of the last launch as the control: that row must show differences)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import two_process_shard as T
import bodyslam_amd.zoedepth as ZD
from bodyslam_amd.zoedepth import ZoeDepthEngine, _ZoePlan
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg_p, wz, wp, frames = T._case()
eng = ZoeDepthEngine(wz, cfg_p, target_hw=T.TARGET, precision="accurate", class_modes="wmean", attn_mode=os.environ.get("ATTN", "single"), neck_mode="full")
cfg = ZD.ZoeConfig()
big = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision="accurate", class_modes="wmean", attn_mode=os.environ.get("ATTN", "single"), neck_mode="full")
if os.environ.get("VICTIM") == "full":                  # the full-size network's kernels (pipelined attention, projector level, the large tiles) as victims, B = 2
    plan = _ZoePlan(big, 2, 480, 640, True)
    plan.frames.copy_(torch.from_numpy(make_sequence(2, 480, 640, seed=3)).cuda())
else:
    plan = _ZoePlan(eng, 4, T.H, T.W, True)
    plan.frames.copy_(torch.from_numpy(frames[:4]).cuda())
P = plan.plan
bplan = _ZoePlan(big, 8, 480, 640, True)
bplan.frames.copy_(torch.from_numpy(make_sequence(8, 480, 640, seed=1)).cuda())
bplan.run(None)
torch.cuda.synchronize()
trig = [(f, a) for (f, a), nm in zip(bplan.plan.calls, bplan.plan.names) if not isinstance(f, str) and nm.endswith(".r1")]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
rows, skipped = [], []
by_ptr = {}
for t in P.keep:
    for x in (t if isinstance(t, tuple) else (t,)):
        if isinstance(x, torch.Tensor):
            by_ptr[x.data_ptr()] = x
gemm_keep = [t for t in P.keep if isinstance(t, tuple)]                 # (desc, A, W, out, kw) per bs_gemm launch, in launch order
gi = 0
import re
for k, ((fn, args), name) in enumerate(zip(P.calls, P.names)):
    if isinstance(fn, str):
        continue
    if k in P.gemm_info:
        tens = [x for x in gemm_keep[gi][1:4] if isinstance(x, torch.Tensor)]
        kw = gemm_keep[gi][4]
        tens += [v for v in kw.values() if isinstance(v, torch.Tensor)]
        gi += 1
    else:
        tens = [by_ptr[a] for a in args if isinstance(a, int) and a in by_ptr]
    # one instance of each kind of launch: layer 0, and level 0 (LEVEL=3: the last, finest level -- the same kernels at their largest)
    if re.match(r"l[1-9]\d*\.", name) or re.match(r"(rt|ro|ra|pj|at|fu)[0-2]" if os.environ.get("LEVEL") == "3" else r"(rt|ro|ra|pj|at|fu)[1-9]", name):
        continue
    tens = [t for t in tens if t.numel() * t.element_size() <= (64 << 20)]
    if not tens:
        continue
    for j in range(k + 1):                                # the plan up to and including the victim, in order, on stream A
        f, a = P.calls[j]
        if not isinstance(f, str):
            f(*a, sA.cuda_stream)
    torch.cuda.synchronize()
    base = [t.clone() for t in tens]
    idem = True
    for _ in range(3):                                    # alone: a launch that changes its own output is not a victim this probe can judge
        fn(*args, sA.cuda_stream)
        torch.cuda.synchronize()
        idem = idem and all(torch.equal(t.view(torch.uint8), b.view(torch.uint8)) for t, b in zip(tens, base))
    if not idem:
        skipped.append(name)
        continue
    bad = done = 0
    while done < n:
        for _ in range(20):
            for f, a in trig:
                f(*a, sB.cuda_stream)
        clones = []
        with torch.cuda.stream(sA):
            for _ in range(50):
                fn(*args, sA.cuda_stream)
                clones.append([t.clone() for t in tens])
        torch.cuda.synchronize()
        bad += sum(int(not all(torch.equal(t.view(torch.uint8), b.view(torch.uint8)) for t, b in zip(c, base))) for c in clones)
        done += 50
    rows.append((name, bad, done))
    if bad:
        print(f"  {name}: {bad} of {done} relaunches differ", flush=True)
print(f"{len(rows)} launches relaunched {n} times each beside {len(trig)} bs_rank1_bias launches x 20 per burst on a second stream:")
print("  differ: " + (", ".join(f"{nm} {b}/{d}" for nm, b, d in rows if b) or "none"))
print("  equal every time: " + ", ".join(nm for nm, b, d in rows if not b))
print("  not idempotent, skipped: " + (", ".join(skipped) or "none"))
