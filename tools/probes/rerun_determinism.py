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
dbg = None
if os.environ.get("DBG") == "1":             # diagnostics build (BODYSLAM_HIP_LIB=.../libbodyslam_hip_diag.so): per-pixel intermediates of the last launch
    import ctypes
    from bodyslam_amd import _lib
    lib = _lib.load_library()
    lib.bs_diag_logbinom_buffer.argtypes = [ctypes.c_void_p]
    dbg = torch.zeros(plan.depth_net.numel() * 8, device="cuda")
    assert lib.bs_diag_logbinom_buffer(dbg.data_ptr()) == 0
# NEIGHBOUR_STREAM=1: a full-size ZoeD_NK plan of the SAME process runs on a second stream beside every rerun (kernels of another stream sharing the CUs)
neighbour = None
if os.environ.get("NEIGHBOUR_STREAM") == "1":
    import bodyslam_amd.zoedepth as ZD
    from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
    cfg_b = ZD.ZoeConfig()
    big = ZoeDepthEngine(random_zoedepth_weights(cfg_b, seed=0), cfg_b, precision="accurate", class_modes="wmean", attn_mode="single", neck_mode="full")
    bplan = _ZoePlan(big, int(os.environ.get("NEIGHBOUR_B", "8")), 480, 640, True)
    bplan.frames.copy_(torch.from_numpy(make_sequence(bplan.frames.shape[0], 480, 640, seed=1)).cuda())
    sB = torch.cuda.Stream()

    def neighbour():
        with torch.cuda.stream(sB):
            bplan.run(None)
            bplan.run(None)
    neighbour()
    torch.cuda.synchronize()
base = {}
plan.run(base)
torch.cuda.synchronize()
base_dbg = dbg.clone() if dbg is not None else None
order = list(base)
bad = {}
fresh = os.environ.get("FRESH") == "1"       # a new plan per iteration (new buffers, new descriptors), as the calibration builds them
skip = lambda nm: False
route = None


RAW = os.environ.get("RAW") == "1"           # compare whole buffers (one plan rerun: what no launch writes keeps its bytes from run to run)


def relaunch_last(it):
    """the depth map differed: where, and does the launch that wrote it repeat the difference on the inputs still in the plan's buffers"""
    P = plan.plan
    k = P.names.index("logbinom")
    fn, args = P.calls[k]
    d = (plan.depth_net != base["depth_net"][0]).nonzero()
    ys, xs = d[:, 1], d[:, 2]
    tiles = sorted({(int(b), int(y) // 16, int(x) // 16) for b, y, x in d.tolist()})
    print(f"  iteration {it}: {len(d)} pixels; images {sorted(set(d[:, 0].tolist()))}; rows {int(ys.min())}..{int(ys.max())} cols {int(xs.min())}..{int(xs.max())}; "
          f"16x16 tiles {tiles[:6]}{'...' if len(tiles) > 6 else ''}; first pixels {d[:12].tolist()}", flush=True)
    if dbg is not None:
        a, b_ = base_dbg.view(-1, 8), dbg.view(-1, 8)
        flat = (d[:, 0] * plan.depth_net.shape[1] + d[:, 1]) * plan.depth_net.shape[2] + d[:, 2]
        comp = ["sum x", "sum interp(Eh)", "sum pre-act", "sum act", "pt0", "pt1", "pt2", "pt3"]
        ne = (a[flat] != b_[flat])
        print("  components that differ at those pixels: " + ", ".join(f"{c}: {int(ne[:, j].sum())}" for j, c in enumerate(comp)), flush=True)
        for q in flat[:3].tolist():
            print(f"    pixel {q}: first run {[f'{v:.6g}' for v in a[q].tolist()]}", flush=True)
            print(f"    {' ' * len(str(q))}        now {[f'{v:.6g}' for v in b_[q].tolist()]}", flush=True)
        other = (a != b_).any(1).nonzero().view(-1)
        print(f"  pixels whose intermediates differ: {len(other)} (depth differs in {len(d)})", flush=True)
    st = torch.cuda.current_stream().cuda_stream
    same = 0
    for _ in range(3):
        fn(*args, st)
        torch.cuda.synchronize()
        same += int(torch.equal(plan.depth_net, base["depth_net"][0]))
    print(f"  the launch alone on the buffers as they are, 3 times: {same} equal to the first run's map", flush=True)


def view(name, t, meta):
    """the part of a marked tensor the plan has written: the routed head's half of a bins tensor, the hi16 values of a fused map"""
    global route
    if RAW:
        return t.view(torch.uint8) if t.dtype != torch.float32 else t.view(torch.int32)
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
    if neighbour is not None:
        neighbour()
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
            if name == "depth_net" and not fresh and len(bad[name]) <= 4:
                relaunch_last(it)
            break                      # the first stage that differs in this iteration
print(f"{n} reruns of the tap path: stages that differed first: " + (", ".join(f"{k}: {len(v)}x (e.g. iteration {v[0][0]}: {v[0][1]} elements, max {v[0][2]:.3e})" for k, v in bad.items()) or "none"), flush=True)
# the same through the plain (non-tap) path: only the final map can be compared
ref = None
nbad = 0
for it in range(n):
    if neighbour is not None:
        neighbour()
    plan.run(None)
    torch.cuda.synchronize()
    d = plan.depth_m.clone()
    if ref is None:
        ref = d
    elif not torch.equal(ref, d):
        nbad += 1
print(f"{n} reruns of the plain path: {nbad} final maps differ from the first", flush=True)
