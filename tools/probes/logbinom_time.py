"""Probe: bs_logbinom_depth_ex at the bench's size (NB = 128, 384 x 512 from 192 x 256): (hi16 | hi8 | lo8) input (matrix-core hidden layer for f16) against
the (hi | lo) input (vector path).   python tools/probes/logbinom_time.py"""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bodyslam_amd import _lib as L
from bodyslam_amd._lib import load_library, check, p as ptr, dt as dtc, stream_ptr
L.init(0)
dev = torch.device("cuda:0")
B, He, We = 128, 192, 256
H, W = 2 * He, 2 * We
g = torch.Generator(device=dev).manual_seed(0)
l8 = torch.randn(B, H, W, 64, generator=g, device=dev, dtype=torch.float16)       # (bit patterns only matter for timing)
Eh = torch.randn(B, He, We, 80, generator=g, device=dev)
bins = F.softplus(torch.randn(B, He, We, 128, generator=g, device=dev) * 2)
w0 = torch.randn(2, 40, 32, generator=g, device=dev) * 0.3
w2 = torch.randn(2, 4, 40, generator=g, device=dev)
b2 = torch.randn(2, 4, generator=g, device=dev)
route = (torch.arange(B, device=dev) % 2).to(torch.int32)
d = torch.empty(B, H, W, device=dev)
for flag, name in ((32, "(hi16 | hi8 | lo8) input"), (16, "(hi | lo) input, vector path")):
    f = lambda: check(load_library().bs_logbinom_depth_ex(ptr(l8), ptr(Eh), ptr(bins), ptr(w0), ptr(w2), ptr(b2), None, 40, ptr(route), ptr(d), B, H, W, He, We,
                                                          0.0212, 50.0, dtc(l8) | flag, stream_ptr()), "x")
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
