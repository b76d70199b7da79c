"""Probe: bs_attention_table at the 416 x 512 network input (hp = 26: 600x480 and 1280x1024 frames), NB = 32 images.  With the diag build
(BODYSLAM_HIP_LIB=bodyslam_amd/libbodyslam_hip_diag.so) BS_ATTN_NO_CLS2=1 gives round 5's 14 + 13-wave blocks for comparison."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd import _lib as L
L.init(0)
NB, nh, hp, wp = int(os.environ.get("NB", "32")), 16, int(os.environ.get("HP", "26")), 32
S = hp * wp + 1
Sp = (S + 63) // 64 * 64
g = torch.Generator().manual_seed(0)
q = (torch.randn(NB, nh, Sp, 64, generator=g) * 0.3).half().cuda()
k = (torch.randn(NB, nh, Sp, 64, generator=g) * 0.3).half().cuda()
vt = torch.randn(NB, nh, 64, Sp, generator=g).half().cuda()
q[:, :, S:] = 0; k[:, :, S:] = 0; vt[:, :, :, S:] = 0
tab = torch.randn(nh, (2 * hp - 1) * 63 + 3, generator=g).cuda()
CP = (NB + 255) // 256 * 256
out = torch.zeros(CP + NB * (S - 1), 1024, dtype=torch.float16, device="cuda")
for _ in range(3):
    L.attention_table(q, k, vt, tab, out, NB, nh, hp, wp, Sp, split=0, grouped=CP)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.attention_table(q, k, vt, tab, out, NB, nh, hp, wp, Sp, split=0, grouped=CP)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 50.0
print(f"bs_attention_table hp={hp} NB={NB} [{os.environ.get('BS_ATTN_NO_CLS2', '') and 'round-5 blocks' or 'default'}]: {us:.1f} us, {4.0 * NB * nh * S * S * 64 / us / 1e6:.0f} TFLOP/s, checksum {out.float().sum().item():.6e}")
