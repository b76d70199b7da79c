"""Probe: one 32-frame batch vs two independent 16-frame plans on two streams (depth network only)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd.zoedepth import ZoeDepthEngine, ZoeConfig, _ZoePlan
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
prec = sys.argv[1] if len(sys.argv) > 1 else "accurate"
cfg = ZoeConfig()
eng = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision=prec)
H, W, B = 480, 640, 32
frames = torch.from_numpy(make_sequence(B, H, W, seed=0)).cuda()
full = _ZoePlan(eng, B, H, W, True)
halves = [_ZoePlan(eng, B // 2, H, W, True) for _ in range(2)]
quarters = [_ZoePlan(eng, B // 4, H, W, True) for _ in range(4)]
side = [torch.cuda.Stream() for _ in range(3)]

def run_full():
    full.frames.copy_(frames); full.plan.run()

def run_split(plans):
    main = torch.cuda.current_stream()
    n = len(plans); per = B // n
    ev = torch.cuda.Event(); ev.record(main)
    for i, p in enumerate(plans):
        st = main if i == 0 else side[i - 1]
        with torch.cuda.stream(st):
            if i: st.wait_event(ev)
            p.frames.copy_(frames[i * per:(i + 1) * per]); p.plan.run()
            if i:
                e = torch.cuda.Event(); e.record(st); main.wait_event(e)

for name, fn in (("full B=32", run_full), ("2 x B=16", lambda: run_split(halves)), ("4 x B=8", lambda: run_split(quarters)), ("full B=32", run_full), ("2 x B=16", lambda: run_split(halves))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(f"{prec} {name}: {dt*1e3:.1f} ms/step  {B/dt:.1f} frames/s (depth only)")
d_full = full.depth_m.clone(); run_split(halves); torch.cuda.synchronize()
print("max |full - halves|:", (d_full - torch.cat([halves[0].depth_m, halves[1].depth_m])).abs().max().item())
