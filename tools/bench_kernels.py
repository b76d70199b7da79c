#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernel on the shapes the ZoeD_NK forward launches
(--nb images; the bench's 32-frame batch is --nb 64).  Interleaved rounds in ONE process; prints TFLOP/s per
(shape, tile variant).  Usage on the GPU box:  python tools/bench_kernels.py [--nb 32] [--reps 20]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyslam_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tiles", default="1,9,10", help="tile ids (igemm.hip); + 100 x ablation bits for diagnostics, e.g. 409 = tile 9 without epilogue")
    ap.add_argument("--only", default="")
    ap.add_argument("--variants", default="wcls,wmean,wmean-compactA", help="f8 shapes: which class modes to run")
    a = ap.parse_args()
    L.init(0)
    dev = torch.device("cuda:0")
    NB = a.nb
    S = 769
    dt = torch.float16
    shapes = []
    M = NB * S
    shapes += [("qkv   K1024 N3072", dict(M=M, N=3072, K=1024), None), ("oproj K1024 N1024 f32res", dict(M=M, N=1024, K=1024), "res"),
               ("fc1   K1024 N4096 gelu", dict(M=M, N=4096, K=1024), "gelu"), ("fc1x  K1024 N4096 noact", dict(M=M, N=4096, K=1024), None),
               ("fc2   K4096 N1024 f32res", dict(M=M, N=1024, K=4096), "res")]
    convs = [("conv 256->256 @96x128", 96, 128, 256, 256), ("conv 256->256 @192x256", 192, 256, 256, 256), ("conv 256->128 @192x256", 192, 256, 256, 128),
             ("conv 256->256 @48x64", 48, 64, 256, 256), ("conv 1024->256 @24x32", 24, 32, 1024, 256), ("conv 128->32 @384x512", 384, 512, 128, 32)]
    tiles = [int(t) for t in a.tiles.split(",")]
    results = []
    for name, g, epi in shapes:
        if a.only and a.only not in name:
            continue
        if a.only == "f8":
            continue
        A = torch.randn(g["M"], g["K"], device=dev).to(dt)
        Wt = (torch.randn(g["N"], g["K"], device=dev) / math.sqrt(g["K"])).to(dt)
        bias = torch.randn(g["N"], device=dev)
        if epi == "res":
            out = torch.randn(g["M"], g["N"], device=dev)
            kw = dict(bias=bias, scale=bias, res=out, ldr=g["N"])
        elif epi == "gelu":
            out = torch.empty(g["M"], g["N"], device=dev, dtype=dt)
            kw = dict(bias=bias, act=L.ACT_GELU)
        else:
            out = torch.empty(g["M"], g["N"], device=dev, dtype=dt)
            kw = dict(bias=bias)
        for t in tiles:
            pl = L.Plan()
            pl.gemm(name, A, Wt, out, M=g["M"], N=g["N"], K=g["K"], lda=g["K"], tile=t, **kw)
            results.append((name, t, 2.0 * g["M"] * g["N"] * g["K"], pl.run))
    # accurate mode's backbone GEMMs: A = (hi16 | hi8 | lo8) rows, W = [W_hi16 | W_lo8 | W_hi8], activation-rounding correction on
    # the first (cls) tile only -- the shapes of the grouped token layout (256 cls / padding rows + NB x 768 patch rows)
    if not a.only or "f8" in a.only:
        MT = 256 + NB * 768
        for name, N_, K_, epi in (("f8 qkv  K1024 N3072", 3072, 1024, "qkv"), ("f8 qkv-shape plain 16-bit out K1024 N3072", 3072, 1024, "plain16"),
                                  ("f8 fc1  K1024 N4096 gelu pair-out lo8 on cls tile only", 4096, 1024, "gelu8lo"), ("f8 o    K1024 N1024 f32res", 1024, 1024, "res"),
                                  ("f8 fc1  K1024 N4096 gelu pair-out", 4096, 1024, "gelu8"), ("f8 fc2  K4096 N1024 f32res", 1024, 4096, "res")):
            A32 = torch.randn(MT, K_, device=dev)
            A8 = torch.empty(MT, 2 * K_, device=dev, dtype=dt)
            L.cast_split(A32, A8, MT, K_, f8=True)
            del A32
            w8, (sb0, sb1) = L.f8_weight(torch.randn(N_, K_) / math.sqrt(K_), dt)
            w8 = w8.to(dev)
            bias = torch.randn(N_, device=dev)
            kw = dict(M=MT, N=N_, K=K_, lda=2 * K_, f8_seg=2 * K_, f8_wonly_from=256, bias=bias,
                      f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1))
            if epi == "res":
                out = torch.randn(MT, N_, device=dev)
                kw.update(scale=bias, res=out, ldr=N_)
            elif epi in ("gelu8", "gelu8lo"):
                out = torch.empty(MT, 2 * N_, device=dev, dtype=dt)
                kw.update(act=L.ACT_GELU, ldo=2 * N_, out_split_off=N_, out_f8=(L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP),
                          out_lo8_rows=256 if epi == "gelu8lo" else 0)
            elif epi == "plain16":      # diagnostics: the QKV product with a plain row-major 16-bit output (no head scatter, no V^T)
                out = torch.empty(MT, N_, device=dev, dtype=dt)
            else:
                Sp = 832
                out = torch.zeros(NB, 16, Sp, 64, device=dev, dtype=dt)
                k2, vt2 = torch.zeros_like(out), torch.zeros(NB, 16, 64, Sp, device=dev, dtype=dt)
                kw.update(qkv=(1024, 769, Sp, 0.18, k2, vt2, True, NB, 256))
            b2 = torch.randn(NB, N_, device=dev)
            for variant in [v for v in ("wcls", "wmean", "wmean-compactA") if v in a.variants.split(",")]:       # 1.5 passes / one pass + per-image bias2 on the patch tiles (the default)
                kv = dict(kw)
                if variant == "wmean-compactA":      # timing experiment only: rows of K 16-bit values, no room for the planes
                    kv.update(lda=K_)
                if variant != "wcls":
                    kv.pop("f8_wonly_from")
                    kv.update(f8_skip_from=256, bias2=(b2, 256, 768))
                    if epi == "gelu8lo":
                        kv.update(out_planes_rows=256)
                for t in tiles:
                    if t % 100 not in (0, 9, 1):
                        continue
                    pl = L.Plan()
                    pl.gemm(name, A8, w8, out, tile=t, **kv)
                    results.append((f"{name} [{variant}]", t, 2.0 * MT * N_ * K_, pl.run))
    for name, H, W_, Ci, Co in convs:
        if a.only and a.only not in name:
            continue
        x = torch.randn(NB, H, W_, Ci, device=dev).to(dt)
        w = (torch.randn(Co, 9 * Ci, device=dev) / math.sqrt(9 * Ci)).to(dt)
        bias = torch.randn(Co, device=dev)
        out = torch.empty(NB, H, W_, Co, device=dev, dtype=dt)
        geom = L.conv_geom(H, W_, Ci, 3, 3, 1, 1)
        for t in ([x_ for x_ in tiles if not (x_ == 6 and Co < 256)] if Co >= 128 else [0]):
            pl = L.Plan()
            pl.gemm(name, x, w, out, M=NB * H * W_, N=Co, K=9 * Ci, lda=Ci, conv=geom, bias=bias, act=L.ACT_RELU, tile=t)
            results.append((name, t, 2.0 * NB * H * W_ * Co * 9 * Ci, pl.run))
    if not a.only or "attn" in a.only:
        Sp = 832
        q = torch.randn(NB, 16, Sp, 64, device=dev).to(dt) * 0.2
        k = torch.randn(NB, 16, Sp, 64, device=dev).to(dt)
        vt = torch.randn(NB, 16, 64, Sp, device=dev).to(dt)
        bias = torch.randn(16, Sp, Sp, device=dev)
        bias[:, :, S:] = -1e30
        ao = torch.empty(NB * S, 1024, device=dev, dtype=dt)
        pl = L.Plan()
        pl.add("attn", "bs_attention", q, k, vt, bias, ao, NB, 16, S, Sp, L.dt(q))
        results.append(("attn  769 tokens x16 heads", 0, 4.0 * NB * 16 * S * S * 64, pl.run))
        tab = torch.randn(16, 47 * 63 + 3, device=dev)
        for split in (0, 32):
            ao2 = torch.empty(NB * S, 1024 * (2 if split else 1), device=dev, dtype=dt)
            pl = L.Plan()
            pl.add("attn_tab", "bs_attention_table", q, k, vt, tab, ao2, NB, 16, 24, 32, Sp, 0, L.dt(q) | split)
            results.append((f"attn_table 24x32 window split{split}", 0, 4.0 * NB * 16 * S * S * 64, pl.run))
    # warm up, then interleaved rounds
    for _, _, _, fn in results:
        fn()
    torch.cuda.synchronize()
    times = [0.0] * len(results)
    for r in range(a.reps):
        for i, (_, _, _, fn) in enumerate(results):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            times[i] += e0.elapsed_time(e1)
    for i, (name, t, fl, _) in enumerate(results):
        ms = times[i] / a.reps
        print(f"{name:28s} tile{t}: {ms * 1e3:9.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
