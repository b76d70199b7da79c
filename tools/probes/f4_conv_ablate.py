#!/usr/bin/env python3
"""Probe: where does an FP4 correction stage spend its time?  3x3 conv 256 -> 256 at 192x256 (the relative head's shape), one 16-bit
pass / FP8 corrections / FP4 corrections, and the FP4 kernel with its scale fetch or its scale reads switched off (ablation bits of
bs_gemm_desc.tile).  Interleaved rounds in one process."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd import _lib as L

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L.init(0)
dev = torch.device("cuda:0")
dt = torch.float16
H, W, C, Co = 192, 256, 256, 256
M = NB * H * W
geom = L.conv_geom(H, W, C, 3, 3, 1, 1)
w = torch.randn(Co, C, 3, 3) / math.sqrt(9 * C)
bias = torch.randn(Co, device=dev)
x32 = torch.randn(M, C, device=dev)
runs = []
# single pass
x16 = x32.to(dt)
w16 = L.conv_weight(w.permute(0, 2, 3, 1)).to(dt).to(dev)
o16 = torch.empty(M, Co, device=dev, dtype=dt)
pl = L.Plan(); pl.gemm("single", x16, w16, o16, M=M, N=Co, K=9 * C, lda=C, conv=geom, bias=bias, act=L.ACT_RELU, tile=9); runs.append(("single pass", pl.run))
# FP8
x8 = torch.empty(M, 2 * C, device=dev, dtype=dt); L.cast_split(x32, x8, M, C, f8=True)
w8, (sb0, sb1) = L.f8_conv_weight(w.permute(0, 2, 3, 1), dt); w8 = w8.to(dev)
o8 = torch.empty(M, 2 * Co, device=dev, dtype=dt)
pl = L.Plan(); pl.gemm("f8", x8, w8, o8, M=M, N=Co, K=9 * C, lda=2 * C, conv=geom, f8_seg=2 * C, f8_scales=(127, sb0, 127 - 11, sb1), bias=bias, act=L.ACT_RELU,
                       ldo=2 * Co, out_split_off=Co, out_f8=(0, 11), tile=9); runs.append(("FP8 corrections, FP8-format out", pl.run))
# FP4
x4 = torch.empty(M, L.f4_pitch(C), device=dev, dtype=dt)
assert L.load_library().bs_cast_split(L.p(x32), L.p(x4), M, C, L.dt(x4) | 64, L.stream_ptr()) == 0
w4, f4 = L.f4_conv_weight(w.permute(0, 2, 3, 1), dt); w4 = w4.to(dev)
o4 = torch.empty(M, L.f4_pitch(Co), device=dev, dtype=dt)
for name, tile, kw in (("FP4 corrections, F4-format out", 9, dict(ldo=L.f4_pitch(Co), out_split_off=Co, out_f4=True)),
                       ("FP4, F8-format out", 9, dict(ldo=L.f4_pitch(Co), out_split_off=Co, out_f8=(0, 11))),
                       ("FP4, no epilogue", 409, dict(ldo=L.f4_pitch(Co), out_split_off=Co, out_f4=True)),
                       ("FP4, no scale fetch, no epilogue", 409 + 3200, dict(ldo=L.f4_pitch(Co), out_split_off=Co, out_f4=True)),
                       ("FP4, no scale fetch/reads, no epilogue", 409 + 9600, dict(ldo=L.f4_pitch(Co), out_split_off=Co, out_f4=True))):
    pl = L.Plan(); pl.gemm(name, x4, w4, o4, M=M, N=Co, K=9 * C, lda=L.f4_pitch(C), conv=geom, f4=f4, bias=bias, act=L.ACT_RELU, tile=tile, **kw)
    runs.append((name, pl.run))
pl = L.Plan(); pl.gemm("f8", x8, w8, o8, M=M, N=Co, K=9 * C, lda=2 * C, conv=geom, f8_seg=2 * C, f8_scales=(127, sb0, 127 - 11, sb1), bias=bias, act=L.ACT_RELU,
                       ldo=2 * Co, out_split_off=Co, out_f8=(0, 11), tile=409); runs.append(("FP8, no epilogue", pl.run))
pl = L.Plan(); pl.gemm("single", x16, w16, o16, M=M, N=Co, K=9 * C, lda=C, conv=geom, bias=bias, act=L.ACT_RELU, tile=409); runs.append(("single pass, no epilogue", pl.run))
for _, fn in runs:
    fn()
torch.cuda.synchronize()
reps = 8
t = [0.0] * len(runs)
for r in range(reps):
    for i, (_, fn) in enumerate(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        t[i] += e0.elapsed_time(e1)
fl = 2.0 * M * Co * 9 * C
for i, (name, _) in enumerate(runs):
    ms = t[i] / reps
    print(f"{name:44s} {ms * 1e3:9.1f} us   {fl / ms / 1e9:7.1f} TFLOP/s algorithmic")
