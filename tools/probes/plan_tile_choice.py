"""Probe: the bench plan's plain GEMM launches outside the backbone, timed on the tile the dispatcher picks and on the 128 x 128 tile (two
blocks per CU: one block's epilogue runs beside the other's main loop -- what a memory-bound small-K product wants).
python tools/probes/plan_tile_choice.py"""
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


gi = 0
tot = [0.0, 0.0]
for i, (fn, args) in enumerate(pl.calls):
    if isinstance(fn, str) or i not in pl.gemm_info:
        continue
    d = pl.keep_descs[gi]
    gi += 1
    name, info = pl.names[i], pl.gemm_info[i]
    if (name.startswith("l") and name[1].isdigit()) or info["tile"] not in (9, 10) or d.N % 128:
        continue
    base = timed(fn, args)
    t0 = d.tile
    d.tile = 1
    alt = timed(fn, args)
    d.tile = t0
    tot[0] += base
    tot[1] += min(base, alt)
    print(f"{name:14s} conv {int(bool(d.conv))} M {d.M:8d} N {d.N:5d} K {d.K:5d} f8_seg {d.f8_seg:5d}: tile {info['tile']:2d} {base:8.1f} us, tile 1 {alt:8.1f} us  {'<-- tile 1' if alt < 0.97 * base else ''}")
print(f"sum {tot[0] / 1e3:.2f} ms -> {tot[1] / 1e3:.2f} ms with the better of the two")
pl.run()
