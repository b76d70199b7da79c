"""Probe: every launch of the bench's ZoeD_NK plan (B = 64, 640x480, accurate mode) timed on its own (20 repetitions between two events),
grouped by entry point and name: where the non-GEMM time of a step sits.   python tools/probes/plan_call_times.py [batch]"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = ZoeConfig()
eng = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision="accurate")
zp = eng.plan_for(B, 480, 640, True)
zp.frames.copy_(torch.from_numpy(make_sequence(B, 480, 640, seed=1)).cuda())
pl = zp.plan
pl.run()
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
rows = []
gi = []
for i, (fn, args) in enumerate(pl.calls):
    if isinstance(fn, str):
        continue
    for _ in range(2):
        fn(*args, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(*args, st)
    e1.record()
    torch.cuda.synchronize()
    rows.append((fn.__name__, pl.names[i], e0.elapsed_time(e1) * 100.0))       # us per launch
    if i in pl.gemm_info:
        gi.append((pl.names[i], e0.elapsed_time(e1) * 100.0, pl.gemm_info[i]))
tot = sum(r[2] for r in rows)
by = collections.defaultdict(lambda: [0, 0.0])
for fn, name, us in rows:
    key = fn if fn != "bs_gemm" else "bs_gemm"
    by[key][0] += 1
    by[key][1] += us
print(f"plan B={B}: {len(rows)} launches, sum of isolated launch times {tot / 1e3:.1f} ms")
for k, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {n:4d} launches {us / 1e3:8.2f} ms  ({us / n:8.1f} us each)  {us / tot * 100:5.1f} %")
print("largest non-GEMM launches:")
for fn, name, us in sorted([r for r in rows if r[0] != "bs_gemm"], key=lambda r: -r[2])[:25]:
    print(f"  {fn:28s} {name:28s} {us:9.1f} us")
print("GEMM launches outside the 24 backbone layers (name, us, tile, algorithmic / executed TFLOP/s):")
neck = [g for g in gi if not (g[0].startswith("l") and g[0][1].isdigit())]
for name, us, info in sorted(neck, key=lambda g: -g[1]):
    print(f"  {name:18s} {us:9.1f} us  tile {info.get('tile')!s:4s} conv {int(bool(info.get('conv')))}  {info.get('alg_flops', 0) / us / 1e6:8.1f} TFLOP/s algorithmic {info.get('flops', 0) / us / 1e6:8.1f} executed")
print(f"  total {sum(g[1] for g in neck) / 1e3:.2f} ms in {len(neck)} launches; backbone {sum(g[1] for g in gi if g not in neck) / 1e3:.2f} ms in {len(gi) - len(neck)}")
