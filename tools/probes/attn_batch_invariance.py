"""Probe: bs_attention_table on a batch vs the same images one at a time, and twice on the same input: bit-equal?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyslam_amd import _lib as L
L.init(0)
dev = torch.device("cuda:0")
hp, wp, nh = 24, 32, 16
S = hp * wp + 1
Sp = (S + 63) // 64 * 64
ntab = (2 * hp - 1) * (2 * wp - 1) + 3
for split in (0, 32, 32 | 64):
    for B in (8, 128):
        g = torch.Generator().manual_seed(3)
        q = torch.zeros(B, nh, Sp, 64, device=dev, dtype=torch.float16); k = torch.zeros_like(q); vt = torch.zeros(B, nh, 64, Sp, device=dev, dtype=torch.float16)
        q[:, :, :S] = (torch.randn(B, nh, S, 64, generator=g) * 0.3).half().to(dev)
        k[:, :, :S] = torch.randn(B, nh, S, 64, generator=g).half().to(dev)
        vt[:, :, :, :S] = torch.randn(B, nh, 64, S, generator=g).half().to(dev)
        tab = torch.randn(nh, ntab, generator=g).to(dev)
        width = nh * 64 * (2 if split else 1)
        out = torch.zeros(B * S, width, device=dev, dtype=torch.float16)
        lib = L.load_library()
        def run(qq, kk, vv, oo, b):
            L.check(lib.bs_attention_table(L.p(qq), L.p(kk), L.p(vv), L.p(tab), L.p(oo), b, nh, hp, wp, Sp, 0, L.dt(qq) | split, L.stream_ptr()), "attn")
        run(q, k, vt, out, B)
        out2 = torch.zeros_like(out)
        run(q, k, vt, out2, B)
        same_rerun = torch.equal(out, out2)
        bad = 0
        for b in (0, 1, B - 1):
            o1 = torch.zeros(S, width, device=dev, dtype=torch.float16)
            run(q[b:b + 1].contiguous(), k[b:b + 1].contiguous(), vt[b:b + 1].contiguous(), o1, 1)
            if not torch.equal(o1, out[b * S:(b + 1) * S]):
                d = (o1.float() - out[b * S:(b + 1) * S].float()).abs()
                bad += 1
                print(f"   image {b}: differs, max {d.max().item():.3e}, rows {(d.amax(1) > 0).sum().item()}")
        print(f"split {split} B {B}: rerun identical {same_rerun}; single-image mismatches {bad}", flush=True)
