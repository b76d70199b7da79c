"""Probe: bs_attention_table at the bench's size (NB = 128, 16 heads, 24 x 32 window) in its build variants, one child process each
(the variant is an environment switch read once): time per launch, rerun determinism over many launches, batch invariance.
    python tools/probes/attn_variants.py            (on the GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {"wpe4 + packed pre-shift, 8-wave blocks + cls_query_pass (default)": {}, "wpe4 + packed, the cls query as a 25th tile (13 + 12-wave blocks)": {"BS_ATTN_NO_CLS2": "1"}, "wpe4, scalar pre-shift (round 3)": {"BS_ATTN_NO_PK": "1"}, "wpe3 + packed": {"BS_ATTN_WPE3": "1"}, "wpe3, scalar": {"BS_ATTN_WPE3": "1", "BS_ATTN_NO_PK": "1"}, "corr (8-wave blocks + cls_query_pass)": {"CORR": "1"}, "corr, the cls query as a 25th tile (4 x 7 waves)": {"CORR": "1", "BS_ATTN_NO_CLS2": "1"}}

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
    corr = os.environ.get("CORR") == "1"
    reruns = int(os.environ.get("RERUNS", "200"))
    g = torch.Generator().manual_seed(3)
    q = torch.zeros(2 * B, nh, Sp, 64, device=dev, dtype=torch.float16)
    k = torch.zeros_like(q)
    vt = torch.zeros(2 * B, nh, 64, Sp, device=dev, dtype=torch.float16)
    for t, shape, sc in ((q, (B, nh, S, 64), 0.3), (k, (B, nh, S, 64), 1.0)):
        v = torch.randn(*shape, generator=g) * sc
        t[:B, :, :S] = v.half().to(dev)
        t[B:, :, :S] = (v - v.half().float()).half().to(dev)
    v = torch.randn(B, nh, 64, S, generator=g)
    vt[:B, :, :, :S] = v.half().to(dev)
    vt[B:, :, :, :S] = (v - v.half().float()).half().to(dev)
    tab = torch.randn(nh, ntab, generator=g).to(dev)
    for split in (32 | 64, 0):
        width = nh * 64 * (2 if split else 1)

        def run(b, out, off=0):
            if corr:
                L.attention_table_corr(q[off:off + b], k[off:off + b], vt[off:off + b], q[B + off:B + off + b], k[B + off:B + off + b],
                                       vt[B + off:B + off + b], tab, out, b, nh, hp, wp, Sp, split=split)
            else:
                L.attention_table(q[off:off + b], k[off:off + b], vt[off:off + b], tab, out, b, nh, hp, wp, Sp, split=split)
        out = torch.zeros(B * S, width, device=dev, dtype=torch.float16)
        run(B, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(B, out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        bad = 0
        o2 = torch.zeros_like(out)
        for _ in range(reruns):
            o2.zero_()
            run(B, o2)
            bad += 0 if torch.equal(o2, out) else 1
        single = 0
        for b in (0, 63, B - 1):
            o1 = torch.zeros(S, width, device=dev, dtype=torch.float16)
            run(1, o1, off=b)
            single += 0 if torch.equal(o1, out[b * S:(b + 1) * S]) else 1
        flops = 4.0 * B * nh * S * S * 64
        print(f"   split {split:3d}: {us:8.1f} us per NB = {B} launch = {flops / us / 1e6:7.1f} TFLOP/s algorithmic; {bad} of {reruns} reruns differ; "
              f"{single} of 3 single-image launches differ", flush=True)
    sys.exit(0)

out = open(os.path.join(ROOT, "gpurun_out", "attn_variants.txt"), "a")
for name, env in VARIANTS.items():
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    txt = f"{name}:\n{r.stdout}{r.stderr[-2000:] if r.returncode else ''}"
    print(txt, flush=True)
    out.write(txt + "\n")
