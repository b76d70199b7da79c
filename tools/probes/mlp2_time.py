"""Probe: bs_mlp2 at the bench's finest level (NB = 128, 192 x 256 pixels), the 128-row two-blocks-per-CU tile against the 256-row tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd import _lib as L
L.init(0)
M, K1, N1, N2 = 128 * 192 * 256, 128, 256, 32
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K1, generator=g).half().cuda()
w1 = (torch.randn(N1, K1, generator=g) / K1 ** 0.5).half().cuda()
w2 = (torch.randn(N2, N1, generator=g) / N1 ** 0.5).half().cuda()
b1, b2 = torch.randn(N1, generator=g).cuda(), torch.randn(N2, generator=g).cuda()
out = torch.empty(M, N2, device="cuda")
for name, ob in (("128-row tile, two blocks per CU", False), ("256-row tile, one block per CU", True), ("128-row tile, two blocks per CU", False)):
    for _ in range(3):
        L.mlp2(x, K1, w1, b1, w2, b2, out, M, K1, N1, N2, one_block=ob)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.mlp2(x, K1, w1, b1, w2, b2, out, M, K1, N1, N2, one_block=ob)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    print(f"bs_mlp2 M = {M} ({name}): {us:.1f} us, {2.0 * M * (K1 * N1 + N1 * N2) / us / 1e6:.0f} TFLOP/s, {M * (K1 * 2 + N2 * 4) / us / 1e3:.0f} GB/s of in + out")
