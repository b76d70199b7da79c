"""Probe: bs_attention_table at the bench's size (NB = 128, 16 heads, 24 x 32 window) with parts of its tile loop switched off (BS_ATTN_ABL, wrong
results -- timing only): what the ring barrier and the wait for the tile DMA cost.   python tools/probes/attn_ablate.py   (one child process per setting)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from bodyslam_amd import _lib as L
    L.init(0)
    dev = torch.device("cuda:0")
    hp, wp, nh, B = 24, 32, 16, 128
    S = hp * wp + 1
    Sp = (S + 63) // 64 * 64
    ntab = (2 * hp - 1) * (2 * wp - 1) + 3
    g = torch.Generator().manual_seed(3)
    q = (torch.randn(B, nh, Sp, 64, generator=g) * 0.3).half().to(dev)
    k = torch.randn(B, nh, Sp, 64, generator=g).half().to(dev)
    vt = torch.randn(B, nh, 64, Sp, generator=g).half().to(dev)
    tab = torch.randn(nh, ntab, generator=g).to(dev)
    for split in (32 | 64, 0):
        out = torch.zeros(B * S, nh * 64 * (2 if split else 1), device=dev, dtype=torch.float16)
        for _ in range(3):
            L.attention_table(q, k, vt, tab, out, B, nh, hp, wp, Sp, split=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.attention_table(q, k, vt, tab, out, B, nh, hp, wp, Sp, split=split)
        e1.record()
        torch.cuda.synchronize()
        print(f"   split {split:3d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
    sys.exit(0)
for name, abl in (("as shipped", 0), ("no ring barrier", 1), ("no DMA wait", 2), ("neither", 3), ("no DMA after tile 1", 4), ("no DMA, no wait, no barrier", 7)):
    print(name, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, BS_ATTN_ABL=str(abl)), check=False)
