import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bodyslam_amd import _lib as L
L.init(0)
LOG2E = 1.4426950408889634
dev = torch.device("cuda:0")
hp, dtype = 24, torch.float16
B, nh, wp = 1, 2, 32
S = hp * wp + 1
Sp = (S + 63) // 64 * 64
ntab = (2 * hp - 1) * (2 * wp - 1) + 3
for amp, pos in ((60.0, 109), (40.0, 109), (30.0, 109), (20.0, 109), (10.0, 109), (60.0, 9), (60.0, 41), (60.0, 64 + 9), (60.0, 300)):
    g = torch.Generator().manual_seed(7)
    q = torch.zeros(B, nh, Sp, 64, device=dev, dtype=dtype); k = torch.zeros_like(q); vt = torch.zeros(B, nh, 64, Sp, device=dev, dtype=dtype)
    qf = torch.randn(B, nh, S, 64, generator=g) * 0.3; kf = torch.randn(B, nh, S, 64, generator=g); vf = torch.randn(B, nh, S, 64, generator=g)
    kf[0, :, pos] = qf[0, :, 5] / qf[0, :, 5].norm(dim=-1, keepdim=True) * amp
    q[:, :, :S] = (qf * LOG2E).to(dtype).to(dev); k[:, :, :S] = kf.to(dtype).to(dev); vt[:, :, :, :S] = vf.transpose(2, 3).to(dtype).to(dev)
    table = torch.randn(nh, ntab, generator=g).to(dev)
    tab2 = (torch.cat([torch.flip(table[:, :ntab - 3], dims=[1]), table[:, ntab - 3:]], 1) * LOG2E).contiguous()
    out = torch.zeros(B * S, nh * 64, device=dev, dtype=dtype)
    L.attention_table(q, k, vt, tab2, out, B, nh, hp, wp, Sp)
    bad = (~torch.isfinite(out.float())).any(1).nonzero().flatten().tolist()
    score = float((q[0, 0, 5].float() * k[0, 0, pos].float()).sum())
    print(f"amp {amp} key {pos}: log2-score of query 5 = {score:.1f}; non-finite rows {bad[:8]}; out[6,:4] = {out[6, :4].tolist()} out[6,64:68] = {out[6, 64:68].tolist()} v = {vf[0, 0, pos, :4].tolist()}", flush=True)
