"""Probe: the relative head's tap-product GEMM (M = NB*192*256, N = 288, K = 128, accurate-mode operands, fp32 output) per tile."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd import _lib as L   # noqa: E402

L.init(0)
dev = torch.device("cuda:0")
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, N, K = NB * 192 * 256, 288, 128
A = torch.zeros(M, 2 * K, device=dev, dtype=torch.float16)
A[:, :K] = torch.randn(4096, K, device=dev).half().repeat(M // 4096, 1)
w = torch.randn(N, K) / K ** 0.5
W8, (sb0, sb1) = L.f8_weight(w, torch.float16)
W8 = W8.to(dev)
out = torch.empty(M, N, device=dev)
kw = dict(M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1))
for tile in (1, 2, 3, 9):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            L.gemm(A, W8, out, tile=tile, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    print(f"tile {tile}: {dt * 1e3:.2f} ms, {(M * (4 * K + 4 * N)) / dt / 1e12:.2f} TB/s of A + out bytes")
out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
for tile in (1, 9):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        L.gemm(A, W8, out16, tile=tile, **kw)
    torch.cuda.synchronize()
    print(f"tile {tile} fp16 out: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
