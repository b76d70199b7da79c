"""Probe: single-frame latency (the reference's own call pattern: one frame per infer_depth_map) vs batch throughput."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd.zoedepth import ZoeDepthEngine, ZoeConfig
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
cfg = ZoeConfig()
for prec in ("accurate", "fast"):
    eng = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision=prec)
    for B in (1, 2, 4, 8):
        frames = torch.from_numpy(make_sequence(B, 480, 640, seed=0)).cuda()
        for _ in range(3): eng.infer(frames)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): eng.infer(frames)
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
        print(f"{prec} B={B}: {t*1e3:.2f} ms/call ({B/t:.1f} frames/s), host issue {t_host*1e3:.2f} ms/call")
        if B <= 2:      # the same through a captured HIP graph
            for _ in range(3): eng.infer(frames, graph=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): eng.infer(frames, graph=True)
            t_host = (time.perf_counter() - t0) / n
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
            print(f"{prec} B={B} hipGraph: {t*1e3:.2f} ms/call ({B/t:.1f} frames/s), host issue {t_host*1e3:.2f} ms/call")
    del eng
