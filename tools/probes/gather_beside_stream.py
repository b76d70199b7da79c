"""Probe: does the last launch of the plan (bs_logbinom_depth_ex) reproduce when OTHER kernels of the SAME process run beside it on a second stream?
(profiles/r06_reproducibility.txt (7): beside a second PROCESS the forms of that kernel that gather the embedding's corners with four- / eight-byte
LDS reads give wrong values in the last 16 lanes of a few waves; alone, never.)  Stream A relaunches the kernel on the untouched buffers of a small
plan and compares every output with the first; stream B runs a full-size ZoeD_NK plan (GEMMs with LDS-DMA, attention, ...) again and again.
    BODYSLAM_HIP_LIB=.../libbodyslam_hip_diag.so BS_LOGBINOM_INTERLEAVED=5 python tools/probes/gather_beside_stream.py [relaunches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import two_process_shard as T
import bodyslam_amd.zoedepth as ZD
from bodyslam_amd.zoedepth import ZoeDepthEngine, _ZoePlan
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg_p, wz, wp, frames = T._case()
eng = ZoeDepthEngine(wz, cfg_p, target_hw=T.TARGET, precision="accurate", class_modes="full", attn_mode="single", neck_mode="full")
plan = _ZoePlan(eng, 4, T.H, T.W, True)
plan.frames.copy_(torch.from_numpy(frames[:4]).cuda())
plan.run(None)
torch.cuda.synchronize()
base = plan.depth_net.clone()
P = plan.plan
k = P.names.index("logbinom")
fn, args = P.calls[k]
# the neighbour: a full-size plan on stream B
cfg = ZD.ZoeConfig()
big = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision="accurate", class_modes="wmean", attn_mode="single", neck_mode="full")
bplan = _ZoePlan(big, int(os.environ.get("NEIGHBOUR_B", "8")), 480, 640, True)
bplan.frames.copy_(torch.from_numpy(make_sequence(bplan.frames.shape[0], 480, 640, seed=1)).cuda())
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sB):
    bplan.run(None)
torch.cuda.synchronize()


only = os.environ.get("NEIGHBOUR_ONLY")          # only the neighbour's launches whose name contains this, repeated to about the whole plan's duration
sub = [(f, a) for (f, a), nm in zip(bplan.plan.calls, bplan.plan.names) if only and not isinstance(f, str) and only in nm and not any(x and x in nm for x in os.environ.get('NEIGHBOUR_NOT', '').split(','))]
if only:
    print(f"neighbour: {len(sub)} launches matching '{only}' x {int(os.environ.get('NEIGHBOUR_REPEAT', '8'))}", flush=True)


def burst(m, with_neighbour):
    bad = 0
    outs = []
    if with_neighbour and only:
        for _ in range(int(os.environ.get("NEIGHBOUR_REPEAT", "8"))):
            for f, a in sub:
                f(*a, sB.cuda_stream)
    elif with_neighbour:
        with torch.cuda.stream(sB):
            bplan.run(None)
    with torch.cuda.stream(sA):
        for _ in range(m):
            fn(*args, sA.cuda_stream)
            outs.append(plan.depth_net.clone())
    torch.cuda.synchronize()
    for o in outs:
        bad += int(not torch.equal(o, base))
    return bad


for label, nb in ((("alone", False), ("beside the neighbour on a second stream", True)) if only else (("alone", False), ("beside the full-size plan on a second stream", True), ("alone", False), ("beside the full-size plan on a second stream", True))):
    bad, done = 0, 0
    while done < n:
        bad += burst(200, nb)
        done += 200
    print(f"{label}: {bad} of {done} relaunches differ from the first", flush=True)
