"""Probe: bs_col_mean + bs_rank1_bias at the backbone's four shapes (NB = 128 groups), microseconds per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd import _lib as L
L.init(0)
dev = torch.device("cuda:0")
G = 128
for name, N, K in (("qkv", 3072, 1024), ("o", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    a = torch.randn(G, K, device=dev).to(torch.bfloat16)
    dw = (torch.randn(N, K, device=dev) * 1e-5).to(torch.bfloat16)
    out = torch.zeros(G, N, device=dev)
    A = torch.randn(256 + G * 768, 2 * K, device=dev).to(torch.float16)
    ab = torch.zeros(G, K, device=dev, dtype=torch.bfloat16)
    for fn, label in ((lambda: L.rank1_bias(a, dw, out), "rank1_bias"), (lambda: L.col_mean(A, 2 * K, 256, 768, G, 8, K, ab, zero=out), "col_mean")):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:4s} N{N} K{K} {label}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
