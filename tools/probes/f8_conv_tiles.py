"""Probe: accurate-mode 3x3 convs of the relative head per tile (NB images of 192x256, (hi16 | hi8 | lo8) pixels in and out)."""
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
H, W = 192, 256
for Ci, Co, tiles in ((256, 128, (1, 2, 9)), (256, 256, (1, 9))):
    x = torch.zeros(NB, H, W, 2 * Ci, device=dev, dtype=torch.float16)
    x[..., :Ci] = torch.randn(H, W, Ci, device=dev).half()
    w = torch.randn(Co, Ci, 3, 3) / (9 * Ci) ** 0.5
    W8, (sb0, sb1) = L.f8_conv_weight(w.permute(0, 2, 3, 1), torch.float16)
    W8 = W8.to(dev)
    bias = torch.zeros(Co, device=dev)
    out = torch.empty(NB, H, W, 2 * Co, device=dev, dtype=torch.float16)
    g = L.conv_geom(H, W, Ci, 3, 3, 1, 1)
    kw = dict(M=NB * H * W, N=Co, K=9 * Ci, lda=2 * Ci, conv=g, f8_seg=2 * Ci, f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1),
              bias=bias, ldo=2 * Co, out_split_off=Co, out_f8=(L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP))
    for tile in tiles:
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                L.gemm(x, W8, out, tile=tile, **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
        print(f"conv {Ci}->{Co} tile {tile}: {dt * 1e3:.2f} ms, {2.0 * NB * H * W * Co * 9 * Ci * 2 / dt / 1e12:.0f} TFLOP/s executed")
