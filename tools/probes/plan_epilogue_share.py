"""Probe: what the epilogues of the bench plan's GEMM launches cost -- every bs_gemm of the ZoeD_NK plan (B = 64, accurate) timed as it is and
with its epilogue ablated (tile + 400: the accumulators are computed and dropped), 10 repetitions each.   python tools/probes/plan_epilogue_share.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd import _lib as L
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine

cfg = ZoeConfig()
eng = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision="accurate")
zp = eng.plan_for(64, 480, 640, True)
zp.frames.copy_(torch.from_numpy(make_sequence(64, 480, 640, seed=1)).cuda())
pl = zp.plan
pl.run()
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
lib = L.load_library()


def timed(fn, args):
    for _ in range(2):
        fn(*args, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(*args, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100.0


rows = []
gi = 0
for i, (fn, args) in enumerate(pl.calls):
    if isinstance(fn, str) or i not in pl.gemm_info:
        continue
    d = pl.keep_descs[gi]
    gi += 1
    full = timed(fn, args)
    t0 = d.tile
    d.tile = (pl.gemm_info[i]["tile"] if t0 % 100 == 0 else t0) + 400
    main = timed(fn, args)
    d.tile = t0
    rows.append((pl.names[i], full, main, bool(pl.gemm_info[i]["conv"])))
torch.cuda.synchronize()
pl.run()          # leave the buffers as a normal run does
bb = [r for r in rows if r[0].startswith("l") and r[0][1].isdigit()]
nk = [r for r in rows if r not in bb]
for label, rs in (("backbone", bb), ("conv-mode launches outside it", [r for r in nk if r[3]]), ("plain launches outside it", [r for r in nk if not r[3]])):
    print(f"{label:32s} {len(rs):3d} launches  {sum(r[1] for r in rs) / 1e3:8.2f} ms, without epilogues {sum(r[2] for r in rs) / 1e3:8.2f} ms")
print("largest epilogues outside the backbone (name, us, us without epilogue):")
for name, full, main, conv in sorted(nk, key=lambda r: -(r[1] - r[2]))[:30]:
    print(f"  {name:16s} {full:9.1f} {main:9.1f}   epilogue {full - main:8.1f} us  ({(full - main) / full * 100:4.1f} %)")
