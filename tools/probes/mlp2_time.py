"""Probe: bs_mlp2 on the finest attractor level's shape (M = 128 x 192 x 256 pixels, pair rows), alone.   python tools/probes/mlp2_time.py [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd import _lib as L
L.init(0)
dev = torch.device("cuda:0")
M, N2 = 128 * 192 * 256, 8
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x = torch.randn(M, 256, device=dev).half()
w1 = (torch.randn(256, 128, device=dev) / 11).half()
w2 = (torch.randn(N2, 256, device=dev) / 16).half()
b1, b2 = torch.randn(256, device=dev), torch.randn(N2, device=dev)
out = torch.empty(M, N2, device=dev)
for _ in range(2):
    L.mlp2(x, 256, w1, b1, w2, b2, out, M, 128, 256, N2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    L.mlp2(x, 256, w1, b1, w2, b2, out, M, 128, 256, N2)
e1.record()
torch.cuda.synchronize()
print(f"bs_mlp2 M={M} N2={N2}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
