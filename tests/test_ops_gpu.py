"""Per-kernel parity of the HIP library (through the C ABI) against plain torch fp32 of the same op.
Every test here needs a real MI355X: run with `pytest -m gpu`."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "ops_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    print(line)


@pytest.fixture(scope="module")
def L():
    from bodyslam_amd import _lib
    _lib.init(0)
    return _lib


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


DT = [torch.float16, torch.bfloat16]


def tol(dtype, k):
    # 16-bit inputs are exact in the fp32 reference; only the fp32-accumulate order and the output rounding differ
    return (2e-3 if dtype == torch.float16 else 1.6e-2)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K,tile", [(128, 128, 64, 1), (300, 256, 192, 1), (257, 64, 128, 2), (1000, 32, 256, 3),
                                        (513, 384, 1024, 11), (64, 8, 64, 0), (2049, 1024, 512, 0), (40000, 256, 256, 0),
                                        (1000, 384, 320, 11), (300, 128, 64, 11), (1000, 640, 448, 9), (5000, 256, 128, 9), (777, 256, 64, 9),
                                        (1000, 640, 448, 10), (5000, 256, 64, 10), (700, 512, 64, 10), (3000, 256, 128, 10), (3000, 1024, 1024, 10)])
def test_gemm_plain(L, dtype, M, N, K, tile):
    A = rnd(M, K, seed=1, dtype=dtype)
    W = rnd(N, K, seed=2, scale=1 / math.sqrt(K), dtype=dtype)
    bias = rnd(N, seed=3)
    out = torch.empty(M, N, device=dev(), dtype=torch.float32)
    L.gemm(A, W, out, M=M, N=N, K=K, lda=K, bias=bias, tile=tile)
    ref = A.float() @ W.float().t() + bias
    err = (out - ref).abs().max().item()
    report(f"gemm_plain {dtype} M{M} N{N} K{K} tile{tile}: max|err|={err:.3e}")
    assert err < 2e-3 * max(1.0, ref.abs().max().item())
    # 16-bit output + relu
    out16 = torch.empty(M, N, device=dev(), dtype=dtype)
    L.gemm(A, W, out16, M=M, N=N, K=K, lda=K, bias=bias, act=L.ACT_RELU, tile=tile)
    assert (out16.float() - F.relu(ref)).abs().max().item() < tol(dtype, K) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", DT)
def test_gemm_epilogue_gelu_scale_residual_groups(L, dtype):
    M, N, K = 2 * 768, 256, 128
    A = rnd(M, K, seed=1, dtype=dtype)
    W = rnd(N, K, seed=2, scale=1 / math.sqrt(K), dtype=dtype)
    bias_g = rnd(2, N, seed=3)
    scale = rnd(N, seed=4)
    # output rows regrouped 768 -> 769 with offset 1 (the patch-embed / readout pattern), fp32 residual in the output geometry
    res = rnd(2 * 769, N, seed=5)
    out = res.clone()
    L.gemm(A, W, out, M=M, N=N, K=K, lda=K, bias=bias_g, bias_group_rows=768, act=L.ACT_GELU, scale=scale, res=out, ldr=N,
           out_group=(768, 769, 1))
    y = F.gelu(A.float() @ W.float().t() + bias_g.repeat_interleave(768, 0)) * scale
    ref = res.clone()
    ref.view(2, 769, N)[:, 1:, :] += y.view(2, 768, N)
    err = (out - ref).abs().max().item()
    report(f"gemm_epilogue {dtype}: max|err|={err:.3e}")
    assert err < 3e-3
    assert torch.equal(out.view(2, 769, N)[:, 0], res.view(2, 769, N)[:, 0])  # untouched cls rows


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,relu_a,tile", [(2, 12, 16, 64, 64, 1, False, 0), (1, 24, 32, 256, 256, 1, True, 0),
                                                                (3, 17, 19, 128, 32, 1, False, 0), (2, 32, 32, 64, 128, 2, False, 0),
                                                                (1, 48, 64, 512, 256, 1, False, 0), (1, 24, 32, 1024, 1024, 2, False, 0),
                                                                (2, 24, 32, 256, 256, 1, True, 11), (2, 24, 32, 64, 256, 1, False, 11),
                                                                (2, 24, 32, 128, 128, 1, True, 11), (2, 31, 33, 64, 128, 2, False, 11),
                                                                (2, 24, 32, 256, 256, 1, True, 9), (3, 17, 19, 128, 256, 1, False, 9),
                                                                (2, 24, 32, 256, 256, 1, True, 10), (3, 17, 19, 128, 256, 1, False, 10),
                                                                (2, 31, 33, 64, 256, 2, False, 10)])
def test_conv3x3(L, dtype, B, H, W, Cin, Cout, stride, relu_a, tile):
    x = rnd(B, H, W, Cin, seed=1, dtype=dtype)                       # NHWC
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(9 * Cin), dtype=dtype)
    bias = rnd(Cout, seed=3)
    wk = L.conv_weight(w.permute(0, 2, 3, 1))                        # [O][I/64][kh][kw][64]
    g = L.conv_geom(H, W, Cin, 3, 3, stride, 1)
    Ho, Wo = g[3], g[4]
    res = rnd(B, Ho, Wo, Cout, seed=4, dtype=dtype)
    out = torch.empty(B, Ho, Wo, Cout, device=dev(), dtype=dtype)
    L.gemm(x, wk, out, M=B * Ho * Wo, N=Cout, K=9 * Cin, lda=Cin, conv=g, relu_a=relu_a, bias=bias, res=res, ldr=Cout, tile=tile)
    xin = x.float().permute(0, 3, 1, 2)
    if relu_a:
        xin = F.relu(xin)
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1).permute(0, 2, 3, 1) + res.float()
    err = (out.float() - ref).abs().max().item()
    report(f"conv3x3 {dtype} B{B} {H}x{W} {Cin}->{Cout} s{stride} relu_a={relu_a} tile{tile}: max|err|={err:.3e}")
    assert err < tol(dtype, 9 * Cin) * max(1.0, ref.abs().max().item())


def _split(x, dtype):
    hi = x.to(dtype)
    lo = (x - hi.float()).to(dtype)
    return hi, lo


def test_gelu_epilogue_accuracy(L):
    """The epilogue's erf-GELU against a float64 evaluation: x passes through an identity GEMM, fp32 output.
    Tolerance 1e-6 absolute (the 16-bit outputs it feeds round at >= 2.4e-4 relative)."""
    M = 4096
    x = torch.linspace(-9.0, 9.0, M * 64, dtype=torch.float64).view(M, 64).to(torch.float16).to(dev())
    eye = torch.eye(64, dtype=torch.float16, device=dev())
    out = torch.empty(M, 64, device=dev())
    L.gemm(x, eye, out, M=M, N=64, K=64, lda=64, act=L.ACT_GELU)
    xd = x.double()
    ref = 0.5 * xd * (1.0 + torch.erf(xd / math.sqrt(2.0)))
    err = (out.double() - ref).abs()
    i = err.argmax()
    report(f"gelu epilogue: max|err|={err.max().item():.3e} at x={xd.flatten()[i].item():.4f}")
    assert err.max().item() < 1e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(256 * 192 + 64, 1024, 128), (256 * 128 + 1, 768, 64), (256 * 64 + 128, 4096, 64)])
def test_gemm_tail_split(L, dtype, M, N, K):
    """Auto tile choice with a ragged last 256-row tile: the full tiles and the <= 128 ragged rows go out as two launches
    (igemm.hip dispatch); every row must still be written, with the fp32-residual epilogue the backbone uses."""
    A = rnd(M, K, seed=1, dtype=dtype)
    W = rnd(N, K, seed=2, scale=1 / math.sqrt(K), dtype=dtype)
    bias = rnd(N, seed=3)
    res = rnd(M, N, seed=4)
    out = res.clone()
    L.gemm(A, W, out, M=M, N=N, K=K, lda=K, bias=bias, res=out, ldr=N)
    ref = A.float() @ W.float().t() + bias + res
    err = (out - ref).abs().max().item()
    report(f"gemm tail split {dtype} M{M} N{N} K{K}: max|err|={err:.3e}")
    assert err < 1e-4 * math.sqrt(K)


@pytest.mark.parametrize("f8", [False, True])
@pytest.mark.parametrize("M,N,K", [(256 * 128 + 128, 1024, 4096), (256 * 64 + 1, 2048, 2048)])
def test_gemm_tail_split_taken(L, f8, M, N, K):
    """The two-launch tail split as the bench batch runs it (o_proj / fc2 at NB = 128: K >= 2048 on the 256x256 tile, the ragged
    rows on a side stream with an event fork / join), plain and with the FP8 correction segment, residual updated in place.
    Proof that the split was taken: the same descriptor with the split switched off (ablate bit 8, one launch) must give the
    same bits, and both must match the fp32 reference."""
    dtype = torch.float16
    A = rnd(M, K, seed=1)
    W = rnd(N, K, seed=2, scale=1 / math.sqrt(K))
    bias = rnd(N, seed=3)
    res = rnd(M, N, seed=4)
    ref = (A.double() @ W.double().t() + bias.double() + res.double())
    if f8:
        w8, (sb0, sb1) = L.f8_weight(W, dtype)
        w8 = w8.to(dev())
        A8 = torch.empty(M, 2 * K, device=dev(), dtype=dtype)
        L.cast_split(A, A8, M, K, f8=True)
        kw = dict(M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1))
        a_, w_ = A8, w8
    else:
        kw = dict(M=M, N=N, K=K, lda=K)
        a_, w_ = A.to(dtype), W.to(dtype)
        ref = a_.double() @ w_.double().t() + bias.double() + res.double()
    assert L.load_library().bs_gemm_tile(L.C.byref(L.make_gemm_desc(a_, w_, res, **kw))) == 9
    out = res.clone()
    L.gemm(a_, w_, out, bias=bias, res=out, ldr=N, **kw)
    out1 = res.clone()
    L.gemm(a_, w_, out1, bias=bias, res=out1, ldr=N, tile=809, **kw)       # tile 9, ablate bit 8: no tail split
    torch.cuda.synchronize()
    err = (out.double() - ref).abs().max().item()
    report(f"gemm tail split taken f8={f8} M{M} N{N} K{K}: max|err|={err:.3e}, split vs single launch identical: {torch.equal(out, out1)}")
    # the ragged rows run on the 128x128 tile in the split and on the 256x256 tile otherwise: same K order, same products -> same bits
    assert torch.equal(out[: M // 256 * 256], out1[: M // 256 * 256])
    assert (out[M // 256 * 256:] - out1[M // 256 * 256:]).abs().max().item() < 1e-4
    assert err < (2e-4 if f8 else 1e-4 * math.sqrt(K))      # split-precision: ~16 bits per operand; single-pass fp16 would be ~1e-3


@pytest.mark.parametrize("tile", [1, 9, 10, 2])
def test_split_precision_plain(L, tile):
    """A = [hi | lo], W' = [W_hi | W_hi | W_lo]: one launch evaluates A_hi W_hi + A_lo W_hi + A_hi W_lo (segments),
    the output is stored as a (hi, lo) pair.  Error vs the fp32 product must be ~1e-6 relative instead of ~5e-4."""
    dtype = torch.float16
    M, N, C = 700, 256 if tile != 2 else 64, 192
    A = rnd(M, C, seed=1)
    W = rnd(N, C, seed=2, scale=1 / math.sqrt(C))
    ah, al = _split(A, dtype)
    wh, wl = _split(W, dtype)
    A2 = torch.cat([ah, al], 1).contiguous()
    W3 = torch.cat([wh, wh, wl], 1).contiguous()
    out = torch.zeros(M, 2 * N, device=dev(), dtype=dtype)
    L.gemm(A2, W3, out, M=M, N=N, K=3 * C, lda=2 * C, seg1=C, ldo=2 * N, out_split_off=N, tile=tile)
    ref = A.double() @ W.double().t()
    got = out[:, :N].double() + out[:, N:].double()
    err = (got - ref).abs().max().item()
    single = torch.empty(M, N, device=dev())
    L.gemm(ah, wh, single, M=M, N=N, K=C, lda=C, tile=tile)
    err1 = (single.double() - ref).abs().max().item()
    report(f"split plain tile{tile}: max|err| split {err:.2e} vs single-pass {err1:.2e}")
    assert err < 2e-5 and err < err1 / 20
    # weights-only split: A single, W' = [W_hi | W_lo]
    W2 = torch.cat([wh, wl], 1).contiguous()
    o32 = torch.empty(M, N, device=dev())
    L.gemm(ah, W2, o32, M=M, N=N, K=2 * C, lda=C, seg1=C, tile=tile)
    refw = ah.double() @ W.double().t()
    assert (o32.double() - refw).abs().max().item() < 2e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [0, 9, 1])
def test_f8_correction_gemm(L, dtype, tile):
    """Split-precision product with the correction passes on the FP8 MFMA: A = [hi16 | hi8 | lo8] (written by LayerNorm's
    `| 32` format), W = [W_hi16 | W_lo8 | W_hi8] (f8_weight).  Must land within a few 1e-5 relative of the fp64 product
    (single pass: ~3e-4 fp16 / 2e-3 bf16), and the fc1-style epilogue must re-emit the same operand format."""
    M, N, K = 700, 512, 256
    x = rnd(M, K, seed=1)                                # fp32 rows; LayerNorm with gamma 1 / beta 0 makes the A operand
    g, b = torch.ones(K, device=dev()), torch.zeros(K, device=dev())
    A8 = torch.empty(M, 2 * K, device=dev(), dtype=dtype)
    check = L.load_library().bs_layernorm(L.p(x), L.p(g), L.p(b), L.p(A8), None, M, K, 1e-6, L.dt(A8) | 32, L.stream_ptr())
    assert check == 0
    xn = F.layer_norm(x.double(), (K,), eps=1e-6)
    w = rnd(N, K, seed=2, scale=1 / math.sqrt(K))
    W8, (sb0, sb1) = L.f8_weight(w, dtype)
    W8 = W8.to(dev())
    out = torch.empty(M, N, device=dev())
    L.gemm(A8, W8, out, M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1), tile=tile)
    ref = xn @ w.double().t()
    err = (out.double() - ref).abs().max().item()
    single = torch.empty(M, N, device=dev())
    L.gemm(A8, w.to(dtype), single, M=M, N=N, K=K, lda=2 * K, tile=tile)
    err1 = (single.double() - ref).abs().max().item()
    report(f"f8 correction gemm {dtype} tile{tile}: max|err|={err:.2e} vs single-pass {err1:.2e}")
    assert err < (1e-4 if dtype == torch.float16 else 8e-4) and err < 0.15 * err1
    # epilogue re-emitting the operand format: out row = [hi16 x N | hi8 x N | lo8 x N]
    o8 = torch.zeros(M, 2 * N, device=dev(), dtype=dtype)
    L.gemm(A8, W8, o8, M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1),
           ldo=2 * N, out_split_off=N, out_f8=(L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP), tile=tile)
    hi = o8[:, :N].float()
    planes = o8[:, N:].contiguous().view(torch.uint8).view(M, 2 * N)
    hi8 = planes[:, :N].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -L.F8_ACT_HI_EXP
    lo8 = planes[:, N:].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -L.F8_ACT_LO_EXP
    assert torch.equal(hi, out.to(dtype).float())
    assert ((hi8 - out).abs() <= out.abs() * 2.0 ** -4 + 2.0 ** -9).all()
    assert ((hi + lo8 - out).abs() <= out.abs() * (2.0 ** -15 if dtype == torch.float16 else 2.0 ** -12) + 2.0 ** -20).all()


@pytest.mark.parametrize("tile,G", [(9, 256), (9, 192), (1, 256)])
def test_gemm_skip_f8_and_group_bias(L, tile, G):
    """bs_gemm f8_skip_from + bias2 (the backbone's "wmean" products): tiles before `skip` evaluate the full split-precision
    product, tiles from `skip` on exactly ONE 16-bit pass plus the per-group bias2 row -- checked against both statements in
    fp64, for groups that coincide with tiles (bias folded per tile) and groups that straddle tiles (per-row lookup)."""
    dtype = torch.float16
    skip, M, N, K = 256, 256 + 3 * 256, 512, 256
    groups = -(-(M - skip) // G)
    x = rnd(M, K, seed=1)
    A8 = to_f8_pairs(x, dtype)
    w = rnd(N, K, seed=2, scale=1 / math.sqrt(K))
    W8, (sb0, sb1) = L.f8_weight(w, dtype)
    bias, b2 = rnd(N, seed=3), rnd(groups, N, seed=4)
    out = torch.empty(M, N, device=dev())
    L.gemm(A8, W8.to(dev()), out, M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1),
           bias=bias, f8_skip_from=skip, bias2=(b2, skip, G), tile=tile)
    xv = from_f8_pairs(A8, K)[1].double()
    full = xv @ w.double().t() + bias.double()
    hi = A8[:, :K].double() @ w.to(dtype).double().t() + bias.double()
    grp = (torch.arange(M - skip, device=dev()) // G)
    hi[skip:] += b2.double()[grp]
    e_head = (out[:skip].double() - full[:skip]).abs().max().item()
    e_tail = (out[skip:].double() - hi[skip:]).abs().max().item()
    report(f"gemm f8_skip_from/bias2 tile{tile} G{G}: cls tile vs split-precision product {e_head:.2e}, patch tiles vs one pass + bias2 {e_tail:.2e}")
    assert e_head < 1e-4 and e_tail < 2e-5           # (the tail differs from its statement by fp32 accumulation order only)
    # the fc2 / o_proj epilogue (fp32 residual + layer scale) and the QKV scatter take the same bias
    res, lam = rnd(M, N, seed=5), rnd(N, seed=6)
    o2 = res.clone()
    L.gemm(A8, W8.to(dev()), o2, M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1),
           bias=bias, scale=lam, res=o2, ldr=N, f8_skip_from=skip, bias2=(b2, skip, G), tile=tile)
    ref2 = res.double() + lam.double() * hi
    assert (o2[skip:].double() - ref2[skip:]).abs().max().item() < 1e-4


def test_col_mean(L):
    dtype = torch.float16
    K, G, rows, row0, step = 512, 5, 96, 256, 8
    A = rnd(row0 + G * rows, 2 * K, seed=7, dtype=dtype)          # pair rows: only the first K columns (the hi16 plane) are averaged
    out = torch.zeros(G, K, device=dev(), dtype=torch.bfloat16)
    junk = torch.ones(G, 300, device=dev())
    L.col_mean(A, 2 * K, row0, rows, G, step, K, out, zero=junk)
    ref = A[row0:, :K].float().view(G, rows, K)[:, ::step].mean(1)
    assert (out.float() - ref).abs().max().item() < 2.0 ** -8 * ref.abs().max().item() + 1e-6
    assert not junk.any()


@pytest.mark.parametrize("G,N,K", [(128, 1024, 1024), (2, 96, 128), (130, 3072, 1024), (128, 1024, 4096)])
def test_rank1_bias(L, G, N, K):
    a = rnd(G, K, seed=8).to(torch.bfloat16)
    dw = (rnd(N, K, seed=9) * 1e-5).to(torch.bfloat16)            # weight-rounding residues are ~2^-12 of the weights
    out = torch.zeros(G, N, device=dev())
    L.rank1_bias(a, dw, out)
    ref = a.double() @ dw.double().t()
    assert (out.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item() + 1e-12
    again = torch.zeros(G, N, device=dev())
    L.rank1_bias(a, dw, again)
    assert torch.equal(out, again)                                # the four k-quarters of a block meet in LDS in wave order: a fixed sum


def to_f8_pairs(x, dtype):
    """fp32 [..., C] -> [..., 2C] `dtype`-typed rows of (hi16 | hi8 | lo8) -- the torch statement of the operand format."""
    import bodyslam_amd._lib as L_
    hi = x.to(dtype)
    lo = x - hi.float()
    hi8 = (x * 2.0 ** L_.F8_ACT_HI_EXP).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    lo8 = (lo * 2.0 ** L_.F8_ACT_LO_EXP).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    C2 = x.shape[-1] * 2
    row = torch.cat([hi.contiguous().view(torch.uint8).view(*x.shape[:-1], C2), hi8, lo8], -1).contiguous()
    return row.view(dtype)


def from_f8_pairs(t, C):
    """[..., 2C] 16-bit rows of (hi16 | hi8 | lo8) -> (hi fp32, value = hi + lo8 * 2^-LO_EXP)."""
    import bodyslam_amd._lib as L_
    hi = t[..., :C].float()
    planes = t[..., C:].contiguous().view(torch.uint8).view(*t.shape[:-1], 2 * C)
    lo = planes[..., C:].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -L_.F8_ACT_LO_EXP
    return hi, hi + lo


@pytest.mark.parametrize("tile", [0, 9, 1])
def test_f8_correction_conv(L, tile):
    """3x3 conv with the correction products on the FP8 MFMA: NHWC pixels [hi16 | hi8 | lo8], weights f8_conv_weight, a residual in
    the same format, output re-emitted in it."""
    dtype = torch.float16
    B, H, Wd, C, Co = 2, 20, 24, 128, 256
    x = rnd(B, H, Wd, C, seed=1)
    w = rnd(Co, C, 3, 3, seed=2, scale=1 / math.sqrt(9 * C))
    res = rnd(B, H, Wd, Co, seed=3)
    x8 = to_f8_pairs(x, dtype)
    r8 = to_f8_pairs(res, dtype)
    W8, (sb0, sb1) = L.f8_conv_weight(w.permute(0, 2, 3, 1), dtype)
    W8 = W8.to(dev())
    g = L.conv_geom(H, Wd, C, 3, 3, 1, 1)
    sc = (127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1)
    out = torch.zeros(B, H, Wd, 2 * Co, device=dev(), dtype=dtype)
    L.gemm(x8, W8, out, M=B * H * Wd, N=Co, K=9 * C, lda=2 * C, conv=g, f8_seg=2 * C, f8_scales=sc, res=r8, ldr=2 * Co, res_f8=True,
           ldo=2 * Co, out_split_off=Co, out_f8=(L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP), tile=tile)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1) + from_f8_pairs(r8, Co)[1].double()
    hi, val = from_f8_pairs(out, Co)
    err = (val.double() - ref).abs().max().item()
    single = torch.empty(B, H, Wd, Co, device=dev())
    L.gemm(x.to(dtype), L.conv_weight(w.permute(0, 2, 3, 1)).to(dtype), single, M=B * H * Wd, N=Co, K=9 * C, lda=C,
           conv=L.conv_geom(H, Wd, C, 3, 3, 1, 1), tile=tile)
    err1 = (single.double() - (ref - from_f8_pairs(r8, Co)[1].double())).abs().max().item()
    report(f"f8 correction conv tile{tile}: max|err|={err:.2e} vs single-pass {err1:.2e}")
    assert err < 2e-4 and err < 0.2 * err1
    # the ReLU'd second output (bs_gemm_desc.out2_relu: what the next residual unit's first convolution reads): relu of the fp32 result in the
    # same row format -- equal to relu of the main output's value up to the lo8 plane's resolution, the main output unchanged by its presence
    out_b, outr = torch.zeros_like(out), torch.zeros_like(out)
    L.gemm(x8, W8, out_b, M=B * H * Wd, N=Co, K=9 * C, lda=2 * C, conv=g, f8_seg=2 * C, f8_scales=sc, res=r8, ldr=2 * Co, res_f8=True,
           ldo=2 * Co, out_split_off=Co, out_f8=(L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP), tile=tile, out_relu=outr)
    assert torch.equal(out_b, out)
    hr, vr = from_f8_pairs(outr, Co)
    assert torch.equal(hr, hi.clamp(min=0)) and (vr >= 0).all()
    assert (vr.double() - val.double().clamp(min=0)).abs().max().item() < 2.0 ** -14
    planes = outr[..., Co:].contiguous().view(torch.uint8).view(B, H, Wd, 2 * Co)
    hi8r = planes[..., :Co].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -L.F8_ACT_HI_EXP
    assert (hi8r - hr).abs().max().item() <= 0.07 * hr.abs().max().item() + 1e-2
    # f8_skip_from = -1 (a product the calibration took down to ONE 16-bit pass): no FP8 stage runs -- the hi16 values of the single-pass product of the
    # hi16 operands -- and out_planes_rows = 256 with it (the producer of a map every consumer of which runs one pass): tiles past row 256 write
    # hi16 only, their plane bytes stay as they were
    M = B * H * Wd
    out_p = torch.full((B, H, Wd, 2 * Co), 7.0, device=dev(), dtype=dtype)
    L.gemm(x8, W8, out_p, M=M, N=Co, K=9 * C, lda=2 * C, conv=g, f8_seg=2 * C, f8_scales=sc, f8_skip_from=-1, f8_wonly_from=-1,
           ldo=2 * Co, out_split_off=Co, out_f8=(L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP), out_lo8_rows=256, out_planes_rows=256, tile=tile)
    xh = x8[..., :C].contiguous()                                                   # the hi16 plane as a plain NHWC map
    w16 = L.conv_weight(w.permute(0, 2, 3, 1)).to(dtype)
    one = torch.empty(B, H, Wd, Co, device=dev())
    L.gemm(xh, w16, one, M=M, N=Co, K=9 * C, lda=C, conv=L.conv_geom(H, Wd, C, 3, 3, 1, 1), tile=tile)
    ref1 = F.conv2d(xh.double().permute(0, 3, 1, 2), w.to(dtype).double(), padding=1).permute(0, 2, 3, 1)
    hp = out_p.view(M, 2 * Co)[:, :Co].float()
    assert (hp.double() - ref1.reshape(M, Co)).abs().max().item() < 2.0 ** -10 * (ref1.abs().max().item() + 1.0)
    assert (hp - one.view(M, Co)).abs().max().item() <= 2.0 ** -10 * (one.abs().max().item() + 1.0)          # (the same product, rounded to 16 bits)
    seven = torch.full((1,), 7.0, dtype=dtype).view(torch.uint8).to(dev())
    first_tile = 256                                                                # rows of the tiles that start below row 256 (128- or 256-row tiles)
    rows_hi_only = out_p.view(M, 2 * Co)[256:].contiguous().view(torch.uint8).view(-1, 4 * Co)
    assert torch.equal(rows_hi_only[:, 2 * Co:], seven.repeat(Co).expand(rows_hi_only.shape[0], 2 * Co))     # neither plane written past row 256
    head = out_p.view(M, 2 * Co)[:first_tile].contiguous().view(torch.uint8).view(-1, 4 * Co)
    assert not torch.equal(head[:, 2 * Co:3 * Co], seven.repeat(Co // 2).expand(first_tile, Co))             # the first tile keeps its hi8 plane


def test_split_precision_conv(L):
    dtype = torch.float16
    B, H, Wd, C, Co = 2, 24, 32, 64, 256
    x = rnd(B, H, Wd, C, seed=1)
    w = rnd(Co, C, 3, 3, seed=2, scale=1 / math.sqrt(9 * C))
    res = rnd(B, H, Wd, Co, seed=3)
    xh, xl = _split(x, dtype)
    rh, rl = _split(res, dtype)
    x2 = torch.cat([xh, xl], -1).contiguous()                       # NHWC with [hi | lo] channels
    r2 = torch.cat([rh, rl], -1).contiguous()
    wk = w.permute(0, 2, 3, 1)                                       # [O][kh][kw][I]
    wh, wl = _split(wk, dtype)
    seg0 = L.conv_weight(torch.cat([wh, wh], -1))                    # segment 0: [W_hi | W_hi] against [hi | lo]
    W3 = torch.cat([seg0, L.conv_weight(wl)], 1).contiguous()        # segment 1: W_lo against hi
    out = torch.zeros(B, H, Wd, 2 * Co, device=dev(), dtype=dtype)
    g = L.conv_geom(H, Wd, 2 * C, 3, 3, 1, 1)
    for tile in (1, 9):
        L.gemm(x2, W3, out, M=B * H * Wd, N=Co, K=9 * 3 * C, lda=2 * C, conv=g, seg1=C, res=r2, ldr=2 * Co, res_split_off=Co,
               ldo=2 * Co, out_split_off=Co, tile=tile)
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1) + res.double()
        got = out[..., :Co].double() + out[..., Co:].double()
        err = (got - ref).abs().max().item()
        report(f"split conv tile{tile}: max|err|={err:.2e}")
        assert err < 3e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("s,C", [(2, 64), (4, 128)])
def test_conv_transpose_shuffle(L, dtype, s, C):
    B, H, W = 2, 6, 8
    x = rnd(B, H, W, C, seed=1, dtype=dtype)
    w = rnd(C, C, s, s, seed=2, scale=1 / math.sqrt(C), dtype=dtype)   # ConvTranspose2d weight [Cin, Cout, k, k]
    bias = rnd(C, seed=3)
    wk = w.permute(2, 3, 1, 0).reshape(s * s * C, C).contiguous()      # [(ky,kx,co)][ci]
    bias_k = bias.repeat(s * s).contiguous()
    out = torch.empty(B, H * s, W * s, C, device=dev(), dtype=dtype)
    L.gemm(x, wk, out, M=B * H * W, N=s * s * C, K=C, lda=C, bias=bias_k, ldo=C, shuffle=(s, C, H, W))
    ref = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.float(), bias, stride=s).permute(0, 2, 3, 1)
    err = (out.float() - ref).abs().max().item()
    report(f"convT {dtype} s{s}: max|err|={err:.3e}")
    assert err < tol(dtype, C) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", DT)
def test_readout_crop_rows(L, dtype):
    """A rows = tokens 1..768 of every image (skip the cls row): conv mode with Hin=1, pad_w=-1."""
    B, T, C, N = 2, 769, 128, 64
    x = rnd(B, T, C, seed=1, dtype=dtype)
    w = rnd(N, C, seed=2, scale=1 / math.sqrt(C), dtype=dtype)
    out = torch.empty(B * (T - 1), N, device=dev(), dtype=torch.float32)
    L.gemm(x, w, out, M=B * (T - 1), N=N, K=C, lda=C, conv=(1, T, C, 1, T - 1, 1, 1, 1, 0, -1))
    ref = (x[:, 1:, :].float() @ w.float().t()).reshape(B * (T - 1), N)
    assert (out - ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,S,nh", [(2, 769, 16), (1, 833, 4), (3, 50, 2)])
def test_qkv_scatter_and_attention(L, dtype, B, S, nh):
    hidden = nh * 64
    Sp = (S + 63) // 64 * 64
    x = rnd(B * S, hidden, seed=1, dtype=dtype)
    wqkv = rnd(3 * hidden, hidden, seed=2, scale=1 / math.sqrt(hidden), dtype=dtype)
    bqkv = rnd(3 * hidden, seed=3, scale=0.1)
    q = torch.zeros(B, nh, Sp, 64, device=dev(), dtype=dtype)
    k = torch.zeros(B, nh, Sp, 64, device=dev(), dtype=dtype)
    vt = torch.zeros(B, nh, 64, Sp, device=dev(), dtype=dtype)
    LOG2E = 1.4426950408889634
    L.gemm(x, wqkv, q, M=B * S, N=3 * hidden, K=hidden, lda=hidden, bias=bqkv, qkv=(hidden, S, Sp, 0.125, k, vt))
    y = (x.float() @ wqkv.float().t() + bqkv).view(B, S, 3, nh, 64)
    qr, kr, vr = y[:, :, 0].permute(0, 2, 1, 3), y[:, :, 1].permute(0, 2, 1, 3), y[:, :, 2].permute(0, 2, 1, 3)
    t = tol(dtype, hidden) * 4
    assert (q[:, :, :S].float() - qr * 0.125).abs().max().item() < t
    assert (k[:, :, :S].float() - kr).abs().max().item() < t
    assert (vt[:, :, :, :S].float() - vr.transpose(2, 3)).abs().max().item() < t
    assert q[:, :, S:].abs().max().item() == 0 and vt[:, :, :, S:].abs().max().item() == 0
    # attention on exactly the 16-bit q/k/v the kernel sees
    bias = torch.full((nh, Sp, Sp), -1.0e30, device=dev())
    bias[:, :S, :S] = rnd(nh, S, S, seed=4)
    bias[:, S:, :S] = 0
    out = torch.empty(B * S, hidden, device=dev(), dtype=dtype)
    # the kernel's contract: scores arrive in the log2 domain (q and bias carry log2(e))
    ql = (q.float() * LOG2E).to(dtype)
    bl = bias.clone()
    bl[:, :S, :S] *= LOG2E
    L.attention(ql, k, vt, bl, out, B, nh, S, Sp)
    qf, kf, vf = ql[:, :, :S].float() / LOG2E, k[:, :, :S].float(), vt[:, :, :, :S].float().transpose(2, 3)
    a = torch.softmax(qf @ kf.transpose(2, 3) + bias[None, :, :S, :S], dim=-1)
    ref = (a @ vf).permute(0, 2, 1, 3).reshape(B * S, hidden)
    err = (out.float() - ref).abs().max().item()
    report(f"attention {dtype} B{B} S{S} nh{nh}: max|err|={err:.3e} (ref max {ref.abs().max().item():.2f})")
    assert err < (4e-3 if dtype == torch.float16 else 3e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,hp,nh,split,grouped", [(2, 24, 16, 0, 0), (1, 26, 4, 0, 0), (3, 3, 2, 0, 0), (2, 24, 2, 16, 0), (2, 24, 2, 32, 0),
                                                   (3, 24, 2, 32, 256), (2, 4, 2, 0, 2),
                                                   # 32 | 64: FP8 planes on the cls query's row only (what the plan asks for); hp 24 / 32: the cls
                                                   # query as cls_query_pass (8-wave blocks), hp 26: as a 27th tile
                                                   (3, 24, 2, 96, 256), (2, 24, 2, 96, 0), (1, 32, 2, 96, 0), (2, 26, 2, 96, 0)])
def test_attention_table(L, dtype, B, hp, nh, split, grouped):
    """bs_attention_table (bias gathered from the per-head table in LDS; Q / K / V^T patches first, cls last) against torch
    softmax attention with HF's gathered [S, S] bias (modeling_beit.py:194-265), for the 24x32 and 26x32 windows of the
    full-size networks and a 3-row toy window; also the pair output formats of accurate mode."""
    from bodyslam_amd.zoedepth import _relative_position_index
    wp = 32
    S = hp * wp + 1
    Sp = (S + 63) // 64 * 64
    hidden = nh * 64
    ntab = (2 * hp - 1) * (2 * wp - 1) + 3
    LOG2E = 1.4426950408889634
    x = rnd(B * S, hidden, seed=1, dtype=dtype)                    # image-major rows, cls first
    wqkv = rnd(3 * hidden, hidden, seed=2, scale=1 / math.sqrt(hidden), dtype=dtype)
    bqkv = rnd(3 * hidden, seed=3, scale=0.1)
    q = torch.zeros(B, nh, Sp, 64, device=dev(), dtype=dtype)
    k = torch.zeros(B, nh, Sp, 64, device=dev(), dtype=dtype)
    vt = torch.zeros(B, nh, 64, Sp, device=dev(), dtype=dtype)
    if grouped:     # grouped rows: the B cls rows, padding up to row `grouped`, then the patch rows image by image
        MT = grouped + B * (S - 1)
        xg = torch.zeros(MT, hidden, device=dev(), dtype=dtype)
        xv = x.view(B, S, hidden)
        xg[:B] = xv[:, 0]
        xg[grouped:] = xv[:, 1:].reshape(-1, hidden)
        L.gemm(xg, wqkv, q, M=MT, N=3 * hidden, K=hidden, lda=hidden, bias=bqkv, qkv=(hidden, S, Sp, 0.125 * LOG2E, k, vt, True, B, grouped))
    else:
        L.gemm(x, wqkv, q, M=B * S, N=3 * hidden, K=hidden, lda=hidden, bias=bqkv, qkv=(hidden, S, Sp, 0.125 * LOG2E, k, vt, True))
    # the scatter put token t at position t-1 and the cls token last
    y = (x.float() @ wqkv.float().t() + bqkv).view(B, S, 3, nh, 64)
    perm = torch.cat([torch.arange(1, S), torch.zeros(1, dtype=torch.long)]).to(dev())
    kr = y[:, :, 1].permute(0, 2, 1, 3)[:, :, perm]
    assert (k[:, :, :S].float() - kr).abs().max().item() < tol(dtype, hidden) * 4
    vr = y[:, :, 2].permute(0, 2, 1, 3)[:, :, perm]
    assert (vt[:, :, :, :S].float() - vr.transpose(2, 3)).abs().max().item() < tol(dtype, hidden) * 4
    table = rnd(nh, ntab, seed=4)                                   # natural-log domain
    tab2 = (torch.cat([torch.flip(table[:, :ntab - 3], dims=[1]), table[:, ntab - 3:]], 1) * LOG2E).contiguous()   # the kernel's operand: body reversed
    mult = 2 if split else 1
    out = torch.zeros((grouped + B * (S - 1)) if grouped else B * S, hidden * mult, device=dev(), dtype=dtype)
    L.attention_table(q, k, vt, tab2, out, B, nh, hp, wp, Sp, split=split, grouped=grouped)
    if grouped:     # back to image-major rows for the comparison
        assert out[B:grouped].abs().max().item() == 0 if grouped > B else True
        out = torch.cat([out[:B].view(B, 1, -1), out[grouped:].view(B, S - 1, -1)], 1).reshape(B * S, -1)
    idx = _relative_position_index(hp, wp).to(dev())
    bias = table[:, idx.view(-1)].view(nh, S, S)                    # [nh, q, k] in token order (cls first)
    inv = torch.empty(S, dtype=torch.long, device=dev())
    inv[perm] = torch.arange(S, device=dev())                       # token -> position
    qf = q[:, :, :S].float()[:, :, inv] / LOG2E
    kf, vf = k[:, :, :S].float()[:, :, inv], vt[:, :, :, :S].float().transpose(2, 3)[:, :, inv]
    a = torch.softmax(qf @ kf.transpose(2, 3) + bias[None], dim=-1)
    ref = (a @ vf).permute(0, 2, 1, 3).reshape(B * S, hidden)
    if split == 16:
        got = out[:, :hidden].float() + out[:, hidden:].float()
    elif split & 32:
        planes = out[:, hidden:].contiguous().view(torch.uint8).view(B * S, 2 * hidden)
        lo8 = planes[:, hidden:].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -L.F8_ACT_LO_EXP
        hi8 = planes[:, :hidden].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -L.F8_ACT_HI_EXP
        if split & 64:      # planes on the cls rows only (image-major rows here: row b * S); the patch rows' plane bytes stay as they were (zero)
            cls_rows = torch.arange(B, device=dev()) * S
            keep = torch.zeros(B * S, 1, device=dev())
            keep[cls_rows] = 1.0
            assert (planes.float() * (1.0 - keep)).abs().max().item() == 0
            got = out[:, :hidden].float() + lo8 * keep
            assert (hi8[cls_rows] - ref[cls_rows]).abs().max().item() < 0.07 * ref.abs().max().item() + 1e-2
        else:
            got = out[:, :hidden].float() + lo8
            assert (hi8 - ref).abs().max().item() < 0.07 * ref.abs().max().item() + 1e-2
    else:
        got = out.float()
    err = (got - ref).abs().max().item()
    report(f"attention_table {dtype} B{B} hp{hp} nh{nh} split{split}: max|err|={err:.3e} (ref max {ref.abs().max().item():.2f})")
    assert err < (4e-3 if dtype == torch.float16 else 3e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,hp,nh,split,grouped,spike", [(2, 24, 4, 0, 0, False), (1, 26, 2, 32, 0, False), (3, 24, 2, 32, 256, False), (2, 4, 2, 0, 2, False),
                                                         (1, 24, 2, 16, 0, True),
                                                         # hp 32 ... 40: ring + table pass 80 KiB of LDS (one block per CU; round-4 advisor:
                                                         # these geometries -- square 512x512 network inputs -- failed at launch)
                                                         (1, 32, 2, 16, 0, False), (1, 40, 1, 32, 0, False)])
def test_attention_table_corr(L, dtype, B, hp, nh, split, grouped, spike):
    """bs_attention_table_corr: the QKV product with qkv_lo_off leaves the rounding residuals of Q / K / V^T behind the values, and the
    split-precision kernel (S = Q K^T + Q_lo K^T + Q K_lo^T, O = V P + V P_lo + V_lo P) reproduces fp64 softmax attention on the
    UNROUNDED q / k / v to ~1e-5 -- two orders below the single-operand kernel on the same inputs.  K carries a few outlier channels
    (what trained BEiT checkpoints put behind their LayerNorms), `spike` adds keys that move the running max."""
    from bodyslam_amd.zoedepth import _relative_position_index
    wp = 32
    S = hp * wp + 1
    Sp = (S + 63) // 64 * 64
    hidden = nh * 64
    ntab = (2 * hp - 1) * (2 * wp - 1) + 3
    LOG2E = 1.4426950408889634
    x = rnd(B * S, hidden, seed=11, dtype=dtype)
    wq = rnd(3 * hidden, hidden, seed=12, scale=1 / math.sqrt(hidden))
    wq[hidden:2 * hidden, :3] *= 12.0                                # K: three input channels weigh 12x (a low-rank outlier part)
    if spike:
        wq[hidden:2 * hidden] *= 3.0
    wqkv = wq.to(dtype)
    bqkv = rnd(3 * hidden, seed=13, scale=0.1)
    QN = B * nh * Sp * 64
    q = torch.zeros(2 * B, nh, Sp, 64, device=dev(), dtype=dtype)
    k = torch.zeros(2 * B, nh, Sp, 64, device=dev(), dtype=dtype)
    vt = torch.zeros(2 * B, nh, 64, Sp, device=dev(), dtype=dtype)
    if grouped:
        MT = grouped + B * (S - 1)
        xg = torch.zeros(MT, hidden, device=dev(), dtype=dtype)
        xv = x.view(B, S, hidden)
        xg[:B] = xv[:, 0]
        xg[grouped:] = xv[:, 1:].reshape(-1, hidden)
        L.gemm(xg, wqkv, q, M=MT, N=3 * hidden, K=hidden, lda=hidden, bias=bqkv, qkv=(hidden, S, Sp, 0.125 * LOG2E, k, vt, True, B, grouped, QN))
    else:
        L.gemm(x, wqkv, q, M=B * S, N=3 * hidden, K=hidden, lda=hidden, bias=bqkv, qkv=(hidden, S, Sp, 0.125 * LOG2E, k, vt, True, 0, 0, QN))
    # hi + lo reproduces the fp32 product far below one 16-bit rounding
    y = (x.double() @ wqkv.double().t() + bqkv.double()).view(B, S, 3, nh, 64)
    perm = torch.cat([torch.arange(1, S), torch.zeros(1, dtype=torch.long)]).to(dev())
    qr = (y[:, :, 0] * (0.125 * LOG2E)).permute(0, 2, 1, 3)[:, :, perm]
    kr = y[:, :, 1].permute(0, 2, 1, 3)[:, :, perm]
    vr = y[:, :, 2].permute(0, 2, 1, 3)[:, :, perm]
    q2, k2, v2 = (q[:B].double() + q[B:].double())[:, :, :S], (k[:B].double() + k[B:].double())[:, :, :S], (vt[:B].double() + vt[B:].double())[:, :, :, :S]
    acc_tol = 4e-6 * max(1.0, kr.abs().max().item()) if dtype == torch.float16 else 1.5e-4 * max(1.0, kr.abs().max().item())
    assert (q2 - qr).abs().max().item() < acc_tol and (k2 - kr).abs().max().item() < acc_tol and (v2 - vr.transpose(2, 3)).abs().max().item() < acc_tol
    assert q[B:, :, S:].abs().max().item() == 0 and vt[B:, :, :, S:].abs().max().item() == 0      # the padding of the residual tensors stays zero
    table = rnd(nh, ntab, seed=14)
    tab2 = (torch.cat([torch.flip(table[:, :ntab - 3], dims=[1]), table[:, ntab - 3:]], 1) * LOG2E).contiguous()
    mult = 2 if split else 1
    rows = (grouped + B * (S - 1)) if grouped else B * S

    def decode(out):
        if grouped:
            out = torch.cat([out[:B].view(B, 1, -1), out[grouped:].view(B, S - 1, -1)], 1).reshape(B * S, -1)
        if split == 16:
            return out[:, :hidden].double() + out[:, hidden:].double()
        if split == 32:
            planes = out[:, hidden:].contiguous().view(torch.uint8).view(B * S, 2 * hidden)
            return out[:, :hidden].double() + planes[:, hidden:].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** -L.F8_ACT_LO_EXP
        return out.double()
    out_c = torch.zeros(rows, hidden * mult, device=dev(), dtype=dtype)
    L.attention_table_corr(q[:B], k[:B], vt[:B], q[B:], k[B:], vt[B:], tab2, out_c, B, nh, hp, wp, Sp, split=split, grouped=grouped)
    out_s = torch.zeros(rows, hidden * mult, device=dev(), dtype=dtype)
    L.attention_table(q[:B], k[:B], vt[:B], tab2, out_s, B, nh, hp, wp, Sp, split=split, grouped=grouped)
    idx = _relative_position_index(hp, wp).to(dev())
    bias = table[:, idx.view(-1)].view(nh, S, S).double()
    inv = torch.empty(S, dtype=torch.long, device=dev())
    inv[perm] = torch.arange(S, device=dev())
    a = torch.softmax((qr[:, :, inv] / LOG2E) @ kr[:, :, inv].transpose(2, 3) + bias[None], dim=-1)      # the UNROUNDED operands, fp64
    ref = (a @ vr[:, :, inv]).permute(0, 2, 1, 3).reshape(B * S, hidden)
    err_c, err_s = (decode(out_c) - ref).abs().max().item(), (decode(out_s) - ref).abs().max().item()
    out_res = {0: 2.0 ** -11, 16: 2.0 ** -21, 32: 2.0 ** -15}[split] * (1.0 if dtype == torch.float16 else 8.0) * max(ref.abs().max().item(), 1.0)
    report(f"attention_table_corr {dtype} B{B} hp{hp} nh{nh} split{split} spike{spike}: max|err| corr {err_c:.3e}, single {err_s:.3e} "
           f"(output resolution {out_res:.1e}, ref max {ref.abs().max().item():.2f}, |k| max {kr.abs().max().item():.1f})")
    assert torch.isfinite(decode(out_c)).all()
    # the corrected kernel is limited by its output format (and, bf16, by 16 bits per operand pair), not by the operands' rounding
    if os.environ.get("BS_TEST_REPORT_ONLY"):
        return
    assert err_c < 2.5 * out_res + (2e-5 if dtype == torch.float16 else 2e-3)
    if split:
        assert err_c < 0.2 * err_s, "the split-precision operands must beat the single 16-bit ones by far"
    out_c2 = torch.zeros_like(out_c)
    L.attention_table_corr(q[:B], k[:B], vt[:B], q[B:], k[B:], vt[B:], tab2, out_c2, B, nh, hp, wp, Sp, split=split, grouped=grouped)
    assert torch.equal(out_c, out_c2)


@pytest.mark.parametrize("hp", [24, 3])
def test_attention_table_running_max_moves_in_both_lane_halves(L, hp):
    """A rare data-dependent branch needs its own test: the deferred running max must move when ONE key far out-scores the rest, in
    whichever lane half holds it (the two halves of a wave hold keys kx & 8 == 0 / != 0 of a 32-key row).  Before round 3 the
    compiler had folded the cross-half maximum away and a spike among the upper half's keys overflowed exp2 to inf; and the shift
    applied in the (wave-uniform) branch must never be negative for a lane that did not ask for it."""
    dtype = torch.float16
    B, nh, wp = 1, 2, 32
    S = hp * wp + 1
    Sp = (S + 63) // 64 * 64
    ntab = (2 * hp - 1) * (2 * wp - 1) + 3
    g = torch.Generator().manual_seed(7)
    q = torch.zeros(B, nh, Sp, 64, device=dev(), dtype=dtype)
    k = torch.zeros(B, nh, Sp, 64, device=dev(), dtype=dtype)
    vt = torch.zeros(B, nh, 64, Sp, device=dev(), dtype=dtype)
    qf = torch.randn(B, nh, S, 64, generator=g) * 0.3
    kf = torch.randn(B, nh, S, 64, generator=g)
    vf = torch.randn(B, nh, S, 64, generator=g)
    # spikes: keys at positions with kx = 8 .. 15 and 24 .. 31 (upper lane half) and one in the lower half, late in the key sequence,
    # aligned with a few queries so that their scores jump by > 40 (log2 domain) over everything before
    # ... and one EARLY in the sequence (key 109 for query 200): every later tile's maximum then lies ~190 below that query's running max,
    # and when another query of the wave moves its max the branch must leave this one alone (exp2(+190) overflowed before the fix)
    for pos, qpos in ((S - 1 - 64 + 9, 5), (S - 1 - 32 + 27, 40), (S - 1 - 32 + 2, 70), (109, 200)):
        if pos < 0 or qpos >= S - 1:
            continue
        kf[0, :, pos] = qf[0, :, qpos] / qf[0, :, qpos].norm(dim=-1, keepdim=True) * 60.0
    q[:, :, :S] = (qf * LOG2E).to(dtype).to(dev())                 # positions: patches first, cls last (the kernel's layout)
    k[:, :, :S] = kf.to(dtype).to(dev())
    vt[:, :, :, :S] = vf.transpose(2, 3).to(dtype).to(dev())
    table = torch.randn(nh, ntab, generator=g).to(dev())
    tab2 = (torch.cat([torch.flip(table[:, :ntab - 3], dims=[1]), table[:, ntab - 3:]], 1) * LOG2E).contiguous()
    out = torch.zeros(B * S, nh * 64, device=dev(), dtype=dtype)
    L.attention_table(q, k, vt, tab2, out, B, nh, hp, wp, Sp)
    assert torch.isfinite(out.float()).all(), "exp2 overflowed: the running max did not follow a spiked key"
    from bodyslam_amd.zoedepth import _relative_position_index
    idx = _relative_position_index(hp, wp).to(dev())
    bias = table[:, idx.view(-1)].view(nh, S, S)                    # token order (cls first)
    perm = torch.cat([torch.arange(1, S), torch.zeros(1, dtype=torch.long)]).to(dev())     # position p holds token perm[p]
    inv = torch.empty(S, dtype=torch.long, device=dev())
    inv[perm] = torch.arange(S, device=dev())
    qt = (q[:, :, :S].double() / LOG2E)[:, :, inv]
    kt, vtok = k[:, :, :S].double()[:, :, inv], vt[:, :, :, :S].double().transpose(2, 3)[:, :, inv]
    a = torch.softmax(qt @ kt.transpose(2, 3) + bias[None].double(), dim=-1)
    ref = (a @ vtok).permute(0, 2, 1, 3).reshape(B * S, nh * 64)
    err = (out.double() - ref).abs().max().item()
    report(f"attention_table spiked keys hp{hp}: max|err|={err:.3e}")
    assert err < 6e-3


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols", [(769 * 2, 1024), (193, 128), (7, 2048)])
def test_layernorm_and_cast(L, dtype, rows, cols):
    x = rnd(rows, cols, seed=1, scale=3.0) + 0.5
    g, b = rnd(cols, seed=2), rnd(cols, seed=3)
    o16 = torch.empty(rows, cols, device=dev(), dtype=dtype)
    o32 = torch.empty(rows, cols, device=dev())
    L.layernorm(x, g, b, o16, o32, rows, cols, 1e-12)
    ref = F.layer_norm(x, (cols,), g, b, 1e-12)
    assert (o32 - ref).abs().max().item() < 2e-5
    assert torch.equal(o16, o32.to(dtype))
    c = torch.empty(rows, cols, device=dev(), dtype=dtype)
    L.cast(x, c)
    assert torch.equal(c, x.to(dtype))


def test_preprocess_matches_oracle(L):
    from oracle import zoedepth_ref as Z
    g = torch.Generator().manual_seed(0)
    for (H, W, nh, nw) in [(480, 640, 384, 512), (480, 600, 416, 512)]:
        f = torch.randint(0, 256, (2, H, W, 3), dtype=torch.uint8, generator=g)
        ref = Z.preprocess(f)
        assert ref.shape[-2:] == (nh, nw)
        fd = f.to(dev())
        out = torch.empty(4, 3, nh, nw, device=dev())
        L.preprocess_image(fd, out, 2, H, W, nh, nw, True)
        e0 = (out[:2].cpu() - ref).abs().max().item()
        e1 = (out[2:].cpu() - torch.flip(ref, dims=[3])).abs().max().item()
        report(f"preprocess {W}x{H}: max|err|={e0:.3e} flipped {e1:.3e}")
        assert e0 < 2e-5 and e1 < 2e-5
        pat = torch.empty(4, (nh // 16) * (nw // 16), 768, device=dev(), dtype=torch.float16)
        L.preprocess_patches(fd, pat, 2, H, W, nh, nw, True)
        refp = F.unfold(out, kernel_size=16, stride=16).transpose(1, 2)   # [4, L, 768] with k = c*256 + ky*16 + kx
        assert torch.equal(pat, refp.to(torch.float16))


@pytest.mark.parametrize("dtype", DT)
def test_resize_and_add(L, dtype):
    B, H, W, C = 2, 12, 16, 64
    x = rnd(B, H, W, C, seed=1, dtype=dtype)
    out = torch.empty(B, 2 * H, 2 * W, C, device=dev(), dtype=dtype)
    L.resize_bilinear_nhwc(x, out, B, H, W, C, 2 * H, 2 * W, True)
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    assert (out.float() - ref).abs().max().item() < tol(dtype, 1) * 4
    y = rnd(B, 2 * H, 2 * W, C, seed=2, dtype=dtype)
    o2 = torch.empty_like(y)
    L.add_resized(y, x, o2, B, H, W, 2 * H, 2 * W, C)
    assert (o2.float() - (ref + y.float())).abs().max().item() < tol(dtype, 1) * 8
    o3 = torch.empty(B, 17, 23, C, device=dev(), dtype=dtype)
    L.resize_bilinear_nhwc(x, o3, B, H, W, C, 17, 23, False)
    ref3 = F.interpolate(x.float().permute(0, 3, 1, 2), size=(17, 23), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    assert (o3.float() - ref3).abs().max().item() < tol(dtype, 1) * 4


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("split", [0, 1, 2])
@pytest.mark.parametrize("geom", [(2, 12, 16, 24, 32, True), (1, 5, 7, 10, 14, True), (1, 6, 5, 12, 10, False), (1, 1, 1, 2, 2, True),
                                  # 32 output channels at x2: the LDS-staged kernel (16 x 16 output tiles, ragged last tiles, both conventions)
                                  (2, 12, 16, 24, 32, True, 32), (1, 41, 53, 82, 106, True, 32), (1, 21, 19, 42, 38, False, 32), (1, 2, 2, 4, 4, True, 32)])
def test_upconv_tapsum(L, dtype, split, geom):
    """relu(conv3x3(interpolate(x))) from low-resolution tap products (the relative head's upsample + conv2, HF
    modeling_zoedepth.py:358-362) against torch's conv2d(interpolate(x)) in fp64: the two orders of the linear steps agree to fp32
    rounding, and the conv's zero padding applies to the UPSAMPLED map (border rows / columns)."""
    B, H, W, Ho, Wo, align = geom[:6]
    C, Co = 16, (geom[6] if len(geom) > 6 else 8)
    x = rnd(B, H, W, C, seed=3).double()
    w = (rnd(Co, C, 3, 3, seed=4) * 0.2).double()
    bias = rnd(Co, seed=5)
    y = torch.einsum("bhwc,ocyx->bhwyxo", x, w).reshape(B, H, W, 9 * Co).float().contiguous()      # n = (ky*3 + kx)*Co + o
    up = F.interpolate(x.permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=align)
    ref = F.relu(F.conv2d(up, w, bias.double(), padding=1)).permute(0, 2, 3, 1)
    out = torch.zeros(B, Ho, Wo, Co * (2 if split else 1), device=dev(), dtype=dtype)
    L.upconv_tapsum(y, bias, out, B, H, W, Co, Ho, Wo, align, split, True)
    if split == 2:
        hi, val = from_f8_pairs(out, Co)
    elif split == 1:
        hi, val = out[..., :Co].float(), out[..., :Co].double() + out[..., Co:].double()
    else:
        hi = val = out.float()
    scale = ref.abs().max().item() + 1.0
    e_hi = (hi.double() - ref).abs().max().item()
    e_val = (val.double() - ref).abs().max().item()
    report(f"upconv_tapsum {dtype} split{split} {geom}: hi err {e_hi:.2e}, value err {e_val:.2e}")
    assert e_hi < tol(dtype, 1) * scale
    if split:
        assert e_val < (2.0 ** -15 if dtype == torch.float16 else 2.0 ** -12) * scale


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("geom", [(2, 12, 20), (1, 24, 32), (3, 5, 7), (6, 48, 64), (1, 2, 2), (2, 9, 33)])
def test_upconv_fused(L, dtype, mode, geom):
    """bs_upconv_fused: relu(conv3x3(interpolate x2(x)) + b), 128 -> 32 channels (the relative head's upsample + conv2, HF
    modeling_zoedepth.py:358-362) in one launch from the low-resolution input, in its three operand modes -- against torch's
    conv2d(interpolate(x)) in fp64 on the values the operands carry, and against the two-launch path it replaces (bs_gemm tap products +
    bs_upconv_tapsum: the same MFMAs, another association of the interpolation's fp32 sum).  Geometries: partial tiles, windows smaller than a
    tile, more tiles than compute units (the persistent loop), 2 x 2 inputs."""
    B, H, W = geom
    C, Co = 128, 32
    x = rnd(B, H, W, C, seed=3)
    w = rnd(Co, C, 3, 3, seed=4) * 0.05
    bias = rnd(Co, seed=5)
    w2 = w.permute(2, 3, 0, 1).reshape(9 * Co, C).contiguous()         # n = (ky*3 + kx)*Co + o
    M = B * H * W
    if mode == 0:
        xin = x.to(dtype).contiguous()
        wp = w2.to(dtype).contiguous()
        scales = (127, 127, 127, 127)
        x_val, w_val = xin.double(), wp.double()
        out = torch.zeros(B, 2 * H, 2 * W, Co, device=dev(), dtype=dtype)
        split = 0
    else:
        xin = torch.empty(B, H, W, 2 * C, device=dev(), dtype=dtype)
        L.cast_split(x.view(M, C), xin, M, C, f8=True)
        wp, (sb0, sb1) = L.f8_weight(w2, dtype)
        wp = wp.to(dev())
        scales = (127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1)
        hi, val = from_f8_pairs(xin, C)
        x_val = (hi if mode == 1 else val).double()                  # mode 1 drops the activation-rounding correction
        w_val = w2.double()
        out = torch.zeros(B, 2 * H, 2 * W, 2 * Co, device=dev(), dtype=dtype)
        split = 2
    L.upconv_fused(xin, wp, bias, out, B, H, W, C, Co, mode=mode, split=split, relu=True, f8_scales=scales)
    up = F.interpolate(x_val.permute(0, 3, 1, 2), size=(2 * H, 2 * W), mode="bilinear", align_corners=True)
    wk = w_val.view(3, 3, Co, C).permute(2, 3, 0, 1)
    ref = F.relu(F.conv2d(up, wk, bias.double(), padding=1)).permute(0, 2, 3, 1)
    # the two-launch path on the same operands
    y9 = torch.empty(M, 9 * Co, device=dev(), dtype=torch.float32)
    if mode == 0:
        L.gemm(xin.view(M, C), wp, y9, M=M, N=9 * Co, K=C, lda=C)
    else:
        L.gemm(xin.view(M, 2 * C), wp, y9, M=M, N=9 * Co, K=C, lda=2 * C, f8_seg=2 * C, f8_wonly_from=-1 if mode == 1 else 0,
               f8_scales=scales)
    two = torch.zeros_like(out)
    L.upconv_tapsum(y9, bias, two, B, H, W, Co, 2 * H, 2 * W, True, split, True)
    dec = (lambda t: from_f8_pairs(t, Co)[1].double()) if split else (lambda t: t.double())
    scale = ref.abs().max().item() + 1.0
    e_ref, e_two = (dec(out) - ref).abs().max().item(), (dec(out) - dec(two)).abs().max().item()
    report(f"upconv_fused {dtype} mode {mode} {geom}: vs fp64 {e_ref:.2e}, vs gemm + tapsum {e_two:.2e} (scale {scale:.2f})")
    assert torch.isfinite(dec(out)).all()
    res = (2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8) if mode == 0 else (2.0 ** -15 if dtype == torch.float16 else 2.0 ** -12)
    # against fp64: the output format's resolution + (modes 1, 2) the e4m3 rounding inside the correction products
    assert e_ref < (1.5 * res + (0.0 if mode == 0 else (3e-5 if dtype == torch.float16 else 3e-4))) * scale
    # against the two-launch path: identical products, the fp32 sums associate differently -- one step of the output format at most
    assert e_two < 1.1 * res * scale
    out2 = torch.zeros_like(out)
    L.upconv_fused(xin, wp, bias, out2, B, H, W, C, Co, mode=mode, split=split, relu=True, f8_scales=scales)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("split", [False, True])
def test_resize_bias_relu(L, dtype, split):
    """bs_resize_bias_relu_nhwc: relu(bilinear x2 (align_corners) of a low-resolution 1x1-convolution output + bias) -- together with that
    convolution it must equal the convolution + ReLU applied to the upsampled map (the bins head's projectors, HF modeling_zoedepth.py:749-772
    on the fusion stage's x2 outputs): a 1x1 convolution commutes with the resize."""
    B, H, W, Cin, C = 2, 9, 13, 32, 64
    x = rnd(B, H, W, Cin, seed=1)
    w = rnd(C, Cin, seed=2, scale=0.3)
    bias = rnd(C, seed=3)
    z = (x.double() @ w.double().t()).float()                                   # the convolution at the low resolution, no bias
    if split:
        zin = torch.empty(B, H, W, 2 * C, device=dev(), dtype=dtype)
        L.cast_split(z.view(-1, C), zin, B * H * W, C)
        zval = zin[..., :C].double() + zin[..., C:].double()
    else:
        zin = z.to(dtype).contiguous()
        zval = zin.double()
    out = torch.zeros(B, 2 * H, 2 * W, C * (2 if split else 1), device=dev(), dtype=dtype)
    L.resize_bias_relu_nhwc(zin, bias, out, B, H, W, C, 2 * H, 2 * W, split=split)
    got = (out[..., :C].double() + out[..., C:].double()) if split else out.double()
    ref = F.relu(F.interpolate(zval.permute(0, 3, 1, 2), size=(2 * H, 2 * W), mode="bilinear", align_corners=True).permute(0, 2, 3, 1) + bias.double())
    scale = ref.abs().max().item() + 1.0
    res = ((2.0 ** -21 if dtype == torch.float16 else 2.0 ** -15) if split else (2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8))
    err = (got - ref).abs().max().item()
    report(f"resize_bias_relu {dtype} split {split}: max|err| {err:.2e} (scale {scale:.2f})")
    assert err < (1.5 * res + 2e-6) * scale and (got >= 0).all()
    # ... and the order of the two linear steps does not matter: conv + bias + relu on the upsampled INPUT
    up = F.interpolate(x.double().permute(0, 3, 1, 2), size=(2 * H, 2 * W), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    ref2 = F.relu(up @ w.double().t() + bias.double())
    assert (got - ref2).abs().max().item() < (1e-5 if (split and dtype == torch.float16) else (2e-3 if dtype == torch.float16 else 2e-2)) * scale      # (the stored z's rounding)


def test_relu_split_f8_without_lo8_plane(L):
    """bs_relu_split with dtype bits 5 | 6: hi16 and hi8 planes as with the lo8 plane, whose bytes are neither read nor written"""
    dtype = torch.float16
    rows, C = 300, 64
    x8 = to_f8_pairs(rnd(rows, C, seed=1), dtype)
    full = torch.zeros(rows, 2 * C, device=dev(), dtype=dtype)
    part = torch.full((rows, 2 * C), 7.0, device=dev(), dtype=dtype)
    from bodyslam_amd._lib import load_library, check, p as ptr, dt as dtc, stream_ptr
    none = torch.full((rows, 2 * C), 7.0, device=dev(), dtype=dtype)
    for out, bits in ((full, 32), (part, 32 | 64), (none, 32 | 128)):
        check(load_library().bs_relu_split(ptr(x8), ptr(out), rows, C, dtc(out) | bits, stream_ptr()), "bs_relu_split")
    fb, pb = full.view(torch.uint8).view(rows, 4 * C), part.view(torch.uint8).view(rows, 4 * C)
    assert torch.equal(fb[:, :3 * C], pb[:, :3 * C]) and (full[:, :C].float() >= 0).all()
    seven = torch.full((1,), 7.0, dtype=dtype).view(torch.uint8).to(dev())
    assert torch.equal(pb[:, 3 * C:], seven.repeat(C // 2).expand(rows, C))
    # bits 5 | 7: hi16 alone (the only reader runs one 16-bit pass)
    nb = none.view(torch.uint8).view(rows, 4 * C)
    assert torch.equal(nb[:, :2 * C], fb[:, :2 * C]) and torch.equal(nb[:, 2 * C:], seven.repeat(C).expand(rows, 2 * C))


def test_resize_f8_without_lo8_plane(L):
    """bs_resize_bilinear_nhwc, flag bit 3: the (hi16 | hi8 | -) output of a map whose every consumer is weight-only -- hi16 and hi8 planes as with the
    plane, the lo8 bytes left untouched"""
    dtype = torch.float16
    B, H, W, C = 2, 6, 8, 64
    x = rnd(B, H, W, C, seed=1)
    x8 = to_f8_pairs(x, dtype)
    full = torch.zeros(B, 2 * H, 2 * W, 2 * C, device=dev(), dtype=dtype)
    part = torch.full((B, 2 * H, 2 * W, 2 * C), 7.0, device=dev(), dtype=dtype)
    from bodyslam_amd._lib import load_library, check, p as ptr, dt as dtc, stream_ptr
    none = torch.full((B, 2 * H, 2 * W, 2 * C), 7.0, device=dev(), dtype=dtype)
    for out, flags in ((full, 1 | 4), (part, 1 | 4 | 8), (none, 1 | 4 | 8 | 16)):
        check(load_library().bs_resize_bilinear_nhwc(ptr(x8), ptr(out), B, H, W, C, 2 * H, 2 * W, flags, dtc(out), stream_ptr()), "bs_resize_bilinear_nhwc")
    nb_ = none.view(torch.uint8).view(B, 2 * H, 2 * W, 4 * C)                     # flag bit 4: hi16 alone
    assert torch.equal(nb_[..., :2 * C], full.view(torch.uint8).view(B, 2 * H, 2 * W, 4 * C)[..., :2 * C])
    assert torch.equal(nb_[..., 2 * C:], torch.full((1,), 7.0, dtype=dtype).view(torch.uint8).to(dev()).repeat(C).expand(B, 2 * H, 2 * W, 2 * C))
    fb, pb = full.view(torch.uint8).view(B, 2 * H, 2 * W, 4 * C), part.view(torch.uint8).view(B, 2 * H, 2 * W, 4 * C)
    assert torch.equal(fb[..., :3 * C], pb[..., :3 * C])                      # hi16 (2C bytes) and hi8 (C bytes)
    seven = torch.full((1,), 7.0, dtype=dtype).view(torch.uint8).to(dev())
    assert torch.equal(pb[..., 3 * C:], seven.repeat(C // 2).expand(B, 2 * H, 2 * W, C))     # the lo8 bytes: not written


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_split_pointwise(L, dtype):
    """(hi | lo) carriers of accurate mode: cast_split, relu_split, split resize and split add_resized keep ~2x the mantissa."""
    B, H, W, C = 2, 12, 16, 64
    x = rnd(B, H, W, C, seed=1)                                         # fp32 values
    xs = torch.empty(B, H, W, 2 * C, device=dev(), dtype=dtype)
    L.cast_split(x, xs, B * H * W, C)
    hi = x.to(dtype)
    assert torch.equal(xs[..., :C], hi) and torch.equal(xs[..., C:], (x - hi.float()).to(dtype))
    val = lambda t: t[..., :C].double() + t[..., C:].double()
    t2 = tol(dtype, 1) ** 2 * 8                                         # two mantissas' worth
    assert (val(xs) - x.double()).abs().max().item() < t2
    r = torch.empty_like(xs)
    L.relu_split(xs, r, B * H * W, C)
    assert (val(r) - val(xs).clamp(min=0)).abs().max().item() < t2
    assert (r[..., :C].float() >= 0).all()
    up = torch.empty(B, 2 * H, 2 * W, 2 * C, device=dev(), dtype=dtype)
    L.resize_bilinear_nhwc(xs, up, B, H, W, C, 2 * H, 2 * W, True, split=True)
    ref = F.interpolate(val(xs).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    t3 = 4e-6 if dtype == torch.float16 else 2e-4                       # fp32 interpolation arithmetic + the re-split
    assert (val(up) - ref).abs().max().item() < t3
    y = rnd(B, 2 * H, 2 * W, C, seed=2)
    ys = torch.empty_like(up)
    L.cast_split(y, ys, B * 4 * H * W, C)
    o = torch.empty_like(up)
    L.add_resized(ys, xs, o, B, H, W, 2 * H, 2 * W, C, split=True)
    assert (val(o) - (ref + val(ys))).abs().max().item() < 2 * t3


def test_f8_pair_pointwise(L):
    """(hi16 | hi8 | lo8) carriers: cast_split, relu_split and the bilinear resize in that format against the torch statement of
    the format (to_f8_pairs / from_f8_pairs)."""
    dtype = torch.float16
    B, H, W, C = 2, 12, 16, 128
    x = rnd(B, H, W, C, seed=1)
    xs = torch.empty(B, H, W, 2 * C, device=dev(), dtype=dtype)
    L.cast_split(x, xs, B * H * W, C, f8=True)
    assert torch.equal(xs.view(torch.int16), to_f8_pairs(x, dtype).view(torch.int16))           # byte-exact format
    hi, val = from_f8_pairs(xs, C)
    assert (val - x).abs().max().item() < 2.0 ** -14 * x.abs().max().item()
    r = torch.empty_like(xs)
    L.relu_split(xs, r, B * H * W, C, f8=True)
    rhi, rval = from_f8_pairs(r, C)
    assert torch.equal(rval, torch.where(hi > 0, val, torch.zeros_like(val)))
    up = torch.empty(B, 2 * H, 2 * W, 2 * C, device=dev(), dtype=dtype)
    L.resize_bilinear_nhwc(xs, up, B, H, W, C, 2 * H, 2 * W, True, split=2)
    ref = F.interpolate(val.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    uhi, uval = from_f8_pairs(up, C)
    assert (uval - ref).abs().max().item() < 2.0 ** -13 * ref.abs().max().item()
    assert torch.equal(up.view(torch.int16)[..., :C], ref.to(dtype).view(torch.int16)) or (uhi - ref).abs().max().item() < 2e-3


def test_attractor_step(L):
    B, Hp, Wp, H, W = 2, 6, 8, 12, 16
    A = F.softplus(rnd(B, H, W, 32, seed=1, scale=2.0))
    prev = F.softplus(rnd(B, Hp, Wp, 128, seed=2, scale=2.0))
    route = torch.tensor([1, 0], dtype=torch.int32, device=dev())
    out = torch.full((B, H, W, 128), -7.0, device=dev())
    L.attractor_step(A, prev, out, route, B, Hp, Wp, H, W, 2, 64, 16)
    c = F.interpolate(prev.permute(0, 3, 1, 2), (H, W), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    for b in range(B):
        g = int(route[b])
        cg, Ag = c[b, :, :, g * 64:(g + 1) * 64], A[b, :, :, g * 16:(g + 1) * 16]
        dx = Ag.unsqueeze(-1) - cg.unsqueeze(-2)
        ref = cg + (dx / (1 + 300 * dx * dx)).sum(-2) / 16
        assert (out[b, :, :, g * 64:(g + 1) * 64] - ref).abs().max().item() < 1e-5
        assert (out[b, :, :, (1 - g) * 64:(2 - g) * 64] == -7.0).all()
    out2 = torch.empty_like(out)
    L.attractor_step(A, prev, out2, None, B, Hp, Wp, H, W, 2, 64, 16)
    assert torch.isfinite(out2).all()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lfmt", [0, 2])
@pytest.mark.parametrize("geom", [(6, 8), (24, 40)])
def test_logbinom_depth(L, dtype, lfmt, geom):
    """lfmt 2: `last` in the (hi16 | hi8 | lo8) row format the relative head writes (its value = hi16 + lo8 2^-11); the second geometry has
    several 16 x 16 output tiles per image (windows that do not start at the map's origin)"""
    B, He, We = 2, geom[0], geom[1]
    H, W = 2 * He, 2 * We
    last = rnd(B, H, W, 32, seed=1, dtype=dtype)
    last_arg, flag = last, 0
    if lfmt == 2:
        last32 = rnd(B, H, W, 32, seed=1)
        last_arg, flag = to_f8_pairs(last32, dtype), 32
        last = from_f8_pairs(last_arg, 32)[1]
    Eh = rnd(B, He, We, 80, seed=2)
    bins = F.softplus(rnd(B, He, We, 128, seed=3, scale=2.0))
    w0 = rnd(2, 40, 32, seed=4, scale=0.3)
    w2 = rnd(2, 4, 40, seed=5, scale=1.0)
    b2 = rnd(2, 4, seed=6)
    route = torch.tensor([0, 1], dtype=torch.int32, device=dev())
    depth = torch.empty(B, H, W, device=dev())
    from bodyslam_amd._lib import load_library, check, p as ptr, dt as dtc, stream_ptr
    check(load_library().bs_logbinom_depth_ex(ptr(last_arg), ptr(Eh), ptr(bins), ptr(w0), ptr(w2), ptr(b2), None, 40, ptr(route), ptr(depth), B, H, W,
                                              He, We, 0.0212, 50.0, dtc(last_arg) | flag, stream_ptr()), "bs_logbinom_depth_ex")
    up = lambda t: F.interpolate(t.permute(0, 3, 1, 2), (H, W), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    Ehu, bu = up(Eh), up(bins)
    for b in range(B):
        g = int(route[b])
        h = F.gelu(Ehu[b, :, :, g * 40:(g + 1) * 40] + last[b].float() @ w0[g].t())
        pt = F.softplus(h @ w2[g].t() + b2[g])
        p = (pt[..., 0] + 1e-4) / (pt[..., 0] + pt[..., 1] + 2e-4)
        t = (pt[..., 2] + 1e-4) / (pt[..., 2] + pt[..., 3] + 2e-4)
        t = (50.0 - 0.0212) * t + 0.0212
        omp = (1 - p).clamp(1e-4, 1.0)
        p = p.clamp(1e-4, 1.0)
        k = torch.arange(64, device=dev(), dtype=torch.float32)
        n = torch.tensor(63.0, device=dev()) + 1e-7
        kk = k + 1e-7
        lb = n * torch.log(n) - kk * torch.log(kk) - (n - kk) * torch.log(n - kk + 1e-7)
        y = lb + k * torch.log(p)[..., None] + (63 - k) * torch.log(omp)[..., None]
        px = torch.softmax(y / t[..., None], dim=-1)
        ref = (px * bu[b, :, :, g * 64:(g + 1) * 64]).sum(-1)
        err = (depth[b] - ref).abs().max().item()
        report(f"logbinom {dtype} lfmt{lfmt} b{b}: max|err|={err:.3e}")
        assert err < 2e-4


def test_postprocess_matches_oracle(L):
    from oracle import zoedepth_ref as Z
    B, H, W, nh, nw = 2, 480, 640, 384, 512
    d = torch.rand(2 * B, nh, nw, generator=torch.Generator().manual_seed(0)) * 3 + 0.2
    ref = Z.postprocess(d[:B], d[B:], H, W)
    dd = d.to(dev())
    m = torch.empty(B, H, W, device=dev())
    u = torch.empty(B, H, W, device=dev(), dtype=torch.int16)
    L.postprocess_depth(dd, m, u, B, H, W, nh, nw, True)
    err = (m.cpu() - ref).abs().max().item()
    report(f"postprocess: max|err|={err:.3e}")
    assert err < 1e-5
    u16 = u.cpu().numpy().view(np.uint16)
    mm = m.cpu().numpy()
    exp = (np.maximum(mm, 0) * 256.0).astype(np.uint16)   # negative metres (bicubic overshoot of this noise input) clamp to 0;
    bad = np.argwhere(u16 != exp)                          # numpy's own cast of a negative float is undefined behaviour
    if len(bad):
        b0 = tuple(bad[0])
        report(f"postprocess u16 mismatches {len(bad)}: at {b0} got {u16[b0]} expected {exp[b0]} metres {m.cpu().numpy()[b0]!r}")
    assert len(bad) == 0
    ok = ref.numpy() > 0.01
    assert np.abs(u16.astype(np.int32) - Z.to_uint16(ref).astype(np.int32))[ok].max() <= 1


@pytest.mark.parametrize("dtype", DT)
def test_small_attention(L, dtype):
    B, S, nh, D = 3, 193, 4, 128
    qkv = rnd(B * S, 3 * D, seed=1)
    out = torch.empty(B * S, D, device=dev(), dtype=dtype)
    L.small_attention(qkv, out, B, S, nh)
    y = qkv.view(B, S, 3, nh, 32)
    q, k, v = y[:, :, 0].transpose(1, 2), y[:, :, 1].transpose(1, 2), y[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32), -1) @ v).transpose(1, 2).reshape(B * S, D)
    assert (out.float() - ref).abs().max().item() < tol(dtype, 1) * 4
    lg = torch.tensor([[0.1, 0.2, 0, 0], [0.3, 0.3, 0, 0], [0.5, -1.0, 0, 0]], device=dev())
    r = torch.empty(3, dtype=torch.int32, device=dev())
    L.route_argmax(lg, 4, r, 3)
    assert r.tolist() == [1, 0, 0]


# ------------------------------------------------------------------------------------------------
# MPEM pieces
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
def test_cyclepose_im2col(L, dtype):
    from oracle import cyclepose_ref as CP
    H, W = 480, 640
    f = torch.randint(0, 256, (3, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    pairs = torch.tensor([[0, 1], [1, 2]], dtype=torch.int32)
    x = CP.center_crop_pair(f, pairs.long())                                   # [P,6,128,128]
    cols = F.unfold(F.pad(x, (3, 3, 3, 3), mode="reflect"), kernel_size=7)     # [P, 6*49, 16384], k = c*49 + ky*7 + kx
    ref = cols.view(2, 6, 49, 128 * 128).permute(0, 3, 2, 1).reshape(2 * 128 * 128, 294)   # k = (ky*7+kx)*6 + c
    out = torch.full((2 * 128 * 128, 320), 9.0, device=dev(), dtype=dtype)
    L.cyclepose_im2col(f.to(dev()), pairs.to(dev()), out, 2, H, W)
    assert torch.equal(out[:, :294].cpu(), ref.to(dtype))
    assert out[:, 294:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DT)
def test_instnorm_avgpool(L, dtype):
    P, HW, C = 3, 64 * 64, 128
    x = rnd(P, HW, C, seed=1, scale=2.0) + 0.7
    o = torch.empty(P, HW, C, device=dev(), dtype=dtype)
    o32 = torch.empty(P, HW, C, device=dev())
    scratch = torch.empty(P * (HW // 256 + 1) * 2 * C, device=dev())
    L.instnorm_relu_nhwc(x, o, o32, scratch, P, HW, C)
    ref = F.relu(F.instance_norm(x.permute(0, 2, 1).reshape(P, C, 64, 64), eps=1e-5)).reshape(P, C, HW).permute(0, 2, 1)
    assert (o32 - ref).abs().max().item() < 2e-5
    assert torch.equal(o, o32.to(dtype))
    pooled = torch.empty(P, C, device=dev())
    L.avgpool_nhwc(x, pooled, P, HW, C)
    assert (pooled - x.mean(1)).abs().max().item() < 1e-5


def test_cyclepose_head(L):
    from oracle import cyclepose_ref as CP
    P, HW, C = 3, 1024, 256
    w = {k: v.to(dev()) for k, v in CP.synth_weights(3).items()}
    pooled = rnd(P, 512, seed=1).abs()
    x2 = rnd(P, HW, C, seed=2).abs()                                            # NHWC
    ws = w["skip_linear.weight"]
    w_pool = ws[:, :512].contiguous()
    w_x2 = ws[:, 512:].view(7, C, HW).permute(0, 2, 1).contiguous()             # [7][HW][C]
    pose7 = torch.empty(P, 7, device=dev())
    T = torch.empty(P, 16, device=dev())
    scratch = torch.empty(P * 64 * 8, device=dev())
    L.cyclepose_head(pooled, x2, w_pool, w_x2, w["skip_linear.bias"], w["pose_dense.1.weight"], w["pose_dense.1.bias"],
                     w["pose_dense.3.weight"], w["pose_dense.3.bias"], pose7, T, scratch, P, HW, C)
    cat = torch.cat([pooled, x2.permute(0, 2, 1).reshape(P, -1)], dim=1)        # NCHW flatten order
    ref7 = F.linear(cat.double(), ws.double(), w["skip_linear.bias"].double()) + \
        F.linear(F.relu(F.linear(pooled.double(), w["pose_dense.1.weight"].double(), w["pose_dense.1.bias"].double())),
                 w["pose_dense.3.weight"].double(), w["pose_dense.3.bias"].double())
    err = (pose7.double() - ref7).abs().max().item()
    report(f"cyclepose_head: max|err pose7|={err:.3e}")
    assert err < 2e-4
    refT = CP.pose_matrix(pose7.cpu())
    assert (T.view(P, 4, 4).cpu() - refT).abs().max().item() < 2e-6


# ------------------------------------------------------------------------------------------------
# 3DM
# ------------------------------------------------------------------------------------------------
def _bp(L, depth, K, poses=None):
    B, H, W = depth.shape
    d = torch.from_numpy(depth.view(np.int16)).to(dev())
    xyz = torch.zeros(B, H * W, 3, device=dev())
    idx = torch.full((B, H * W), -1, dtype=torch.int32, device=dev())
    cnt = torch.zeros(B, dtype=torch.int32, device=dev())
    scratch = torch.zeros(B * (H * W // 256 + 2), dtype=torch.int32, device=dev())
    pz = None if poses is None else torch.from_numpy(poses.reshape(B, 16)).to(dev())
    L.backproject(d, K, 1000.0, 3.0, pz, xyz, idx, cnt, scratch, B, H, W)
    torch.cuda.synchronize()
    return xyz.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()


def test_backproject_golden_and_random(L, golden_dir):
    from oracle import geom3d_ref as G
    g = np.load(os.path.join(golden_dir, "geom3d_backproject.npz"))
    xyz, idx, cnt = _bp(L, g["depth"][None], tuple(g["K"]))
    m = int(cnt[0])
    assert m == len(g["idx"])
    assert np.array_equal(idx[0, :m], g["idx"])                       # bit-exact indices
    assert np.array_equal(xyz[0, :m], g["xyz"].astype(np.float32))    # and, without a pose, bit-exact points
    # full-size frames, ragged validity, with poses
    rng = np.random.default_rng(1)
    depth = rng.integers(0, 4000, size=(3, 480, 640)).astype(np.uint16)
    depth[1] = 0                                                      # empty frame
    depth[2, :, :] = 1500                                             # fully valid frame
    chain = np.load(os.path.join(golden_dir, "geom3d_chain.npz"))["g_abs"]
    poses = chain[[10, 20, 30]]
    xyz, idx, cnt = _bp(L, depth, G.REF_INTRINSICS, poses)
    for b in range(3):
        rx, ri = G.backproject(depth[b], pose=poses[b])
        assert int(cnt[b]) == len(ri)
        assert np.array_equal(idx[b, :len(ri)], ri)
        assert np.allclose(xyz[b, :len(ri)], rx, rtol=0, atol=1e-6)
    assert cnt[1] == 0 and cnt[2] == 480 * 640
    report(f"backproject: counts {cnt.tolist()} indices exact")


def test_backproject_odd_sizes(L):
    from oracle import geom3d_ref as G
    rng = np.random.default_rng(2)
    for (H, W) in [(1, 1), (3, 5), (37, 61), (1, 2049)]:
        depth = rng.integers(0, 3500, size=(2, H, W)).astype(np.uint16)
        xyz, idx, cnt = _bp(L, depth, G.REF_INTRINSICS)
        for b in range(2):
            rx, ri = G.backproject(depth[b])
            assert int(cnt[b]) == len(ri) and np.array_equal(idx[b, :len(ri)], ri)
            assert np.array_equal(xyz[b, :len(ri)], rx)


def test_pose_chain(L, golden_dir):
    g = np.load(os.path.join(golden_dir, "geom3d_chain.npz"))
    t_rel = torch.from_numpy(g["t_rel"]).to(dev())
    N = t_rel.shape[0]
    out = torch.empty(N + 1, 16, dtype=torch.float64, device=dev())
    L.pose_chain(t_rel.view(N, 16), N, None, out)
    got = out.cpu().numpy().reshape(N + 1, 4, 4)
    err = np.abs(got - g["g_abs"]).max()
    report(f"pose_chain N={N}: max|err| vs reference = {err:.3e}")
    assert err < 1e-9          # fp64 Jacobi SVD vs LAPACK gesdd, accumulated over 1000 steps
    g0 = g["g_abs"][500]
    out2 = torch.empty(11, 16, dtype=torch.float64, device=dev())
    L.pose_chain(t_rel[500:510].reshape(10, 16).contiguous(), 10, g0.reshape(16), out2)
    assert np.abs(out2.cpu().numpy().reshape(11, 4, 4) - g["g_abs"][500:511]).max() < 1e-12
    # reflection input: the det correction must act on the smallest singular direction
    from oracle import geom3d_ref as G
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.diag([1.0, 0.9, -0.8]).astype(np.float32)
    out3 = torch.empty(2, 16, dtype=torch.float64, device=dev())
    L.pose_chain(torch.from_numpy(T.reshape(1, 16)).to(dev()), 1, None, out3)
    assert np.abs(out3.cpu().numpy()[1].reshape(4, 4) - G.pose_chain(T[None])[1]).max() < 1e-12


def test_attention_table_is_deterministic_at_scale(L):
    """The bench's attention launch (128 images, 16 heads, 24 x 32 windows) twice on the same input, and three of its images one at a time:
    the same bits.  (A packed-math variant of the kernel passed every accuracy test at small batches and was wrong now and then at
    this size -- a register hazard that shows only when the chip is full; the B = 64 plan then differed from the B = 1 plan.)"""
    dtype = torch.float16
    hp, wp, nh, B = 24, 32, 16, 128
    S = hp * wp + 1
    Sp = (S + 63) // 64 * 64
    ntab = (2 * hp - 1) * (2 * wp - 1) + 3
    g = torch.Generator().manual_seed(3)
    q = torch.zeros(B, nh, Sp, 64, device=dev(), dtype=dtype)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, nh, 64, Sp, device=dev(), dtype=dtype)
    q[:, :, :S] = (torch.randn(B, nh, S, 64, generator=g) * 0.3).to(dtype).to(dev())
    k[:, :, :S] = torch.randn(B, nh, S, 64, generator=g).to(dtype).to(dev())
    vt[:, :, :, :S] = torch.randn(B, nh, 64, S, generator=g).to(dtype).to(dev())
    tab = torch.randn(nh, ntab, generator=g).to(dev())
    for split in (0, 32):
        width = nh * 64 * (2 if split else 1)
        lib = L.load_library()

        def run(qq, kk, vv, b):
            out = torch.zeros(b * S, width, device=dev(), dtype=dtype)
            L.check(lib.bs_attention_table(L.p(qq), L.p(kk), L.p(vv), L.p(tab), L.p(out), b, nh, hp, wp, Sp, 0, L.dt(qq) | split, L.stream_ptr()), "attn")
            return out
        a = run(q, k, vt, B)
        for _ in range(3):
            assert torch.equal(run(q, k, vt, B), a), "bs_attention_table is not deterministic at B = 128"
        for b in (0, 63, B - 1):
            one = run(q[b:b + 1].contiguous(), k[b:b + 1].contiguous(), vt[b:b + 1].contiguous(), 1)
            assert torch.equal(one, a[b * S:(b + 1) * S]), f"image {b} of the batch differs from the same image alone"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N2,pairs", [(1000, 8, True), (256, 32, False), (777, 16, True), (70000, 8, True), (5, 4, False)])
def test_mlp2(L, dtype, M, N2, pairs):
    """bs_mlp2 (the attractor MLP in one launch, HF modeling_zoedepth.py:665-700) against the two bs_gemm launches it replaces -- bit for bit --
    and against torch in fp64; ragged last block, pair rows (only the hi half is read), every output width."""
    K1, N1 = 128, 256
    ldx = 2 * K1 if pairs else K1
    x = rnd(M, ldx, seed=11, dtype=dtype)
    w1 = (rnd(N1, K1, seed=12) / K1 ** 0.5).to(dtype)
    w2 = (rnd(N2, N1, seed=13) / N1 ** 0.5).to(dtype)
    b1, b2 = rnd(N1, seed=14), rnd(N2, seed=15)
    out = torch.full((M, N2), -7.0, device=dev())
    L.mlp2(x, ldx, w1, b1, w2, b2, out, M, K1, N1, N2, L.ACT_SOFTPLUS_FAST)
    hid = torch.empty(M, N1, device=dev(), dtype=dtype)
    L.gemm(x, w1, hid, M=M, N=N1, K=K1, lda=ldx, bias=b1, act=L.ACT_RELU)
    two = torch.empty(M, N2, device=dev())
    L.gemm(hid, w2, two, M=M, N=N2, K=N1, lda=N1, bias=b2, act=L.ACT_SOFTPLUS_FAST)
    torch.cuda.synchronize()
    assert torch.equal(out, two), f"max |fused - two launches| = {(out - two).abs().max().item():.3e}"
    h64 = torch.relu(x[:, :K1].double() @ w1.double().t() + b1.double()).to(dtype).double()
    ref = F.softplus(h64 @ w2.double().t() + b2.double())
    err = (out.double() - ref).abs().max().item()
    report(f"mlp2 {dtype} M={M} N2={N2} pairs={pairs}: max err vs fp64 {err:.2e}")
    assert err < 5e-3       # (a hidden unit on a 16-bit rounding boundary may round the other way than in fp64)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(2, 6, 8, 12, 16, 8, True), (1, 12, 16, 24, 32, 16, True), (3, 5, 7, 9, 13, 32, False), (1, 1, 1, 1, 1, 8, True),
                                  (1, 24, 32, 48, 64, 8, True)])
def test_mlp2_add(L, dtype, geom):
    """bs_mlp2_add (an attractor level in one launch) against bs_add_resized followed by bs_mlp2 on its output -- bit for bit: ragged last
    block, pair and single rows, 1x1 maps, every output width."""
    B, Hp, Wp, H, W, N2, pairs = geom
    K1, N1 = 128, 256
    m2 = 2 if pairs else 1
    emb = rnd(B, H, W, K1 * m2, seed=21, dtype=dtype)
    prev = rnd(B, Hp, Wp, K1 * m2, seed=22, dtype=dtype)
    if pairs:       # lo halves of a realistic magnitude
        emb[..., K1:] *= 2.0 ** -11
        prev[..., K1:] *= 2.0 ** -11
    w1 = (rnd(N1, K1, seed=23) / K1 ** 0.5).to(dtype)
    w2 = (rnd(N2, N1, seed=24) / N1 ** 0.5).to(dtype)
    b1, b2 = rnd(N1, seed=25), rnd(N2, seed=26)
    M = B * H * W
    out = torch.full((M, N2), -7.0, device=dev())
    L.mlp2_add(emb, prev, w1, b1, w2, b2, out, B, Hp, Wp, H, W, K1, N1, N2, L.ACT_SOFTPLUS_FAST, split=pairs)
    y = torch.empty_like(emb)
    L.add_resized(emb, prev, y, B, Hp, Wp, H, W, K1, split=pairs)
    two = torch.empty(M, N2, device=dev())
    L.mlp2(y, K1 * m2, w1, b1, w2, b2, two, M, K1, N1, N2, L.ACT_SOFTPLUS_FAST)
    torch.cuda.synchronize()
    assert torch.equal(out, two), f"max |fused - two launches| = {(out - two).abs().max().item():.3e}"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(2, 6, 16, True), (1, 12, 32, False), (3, 5, 64, True), (2, 96, 128, True), (2, 96, 128, False)])
def test_projector_level(L, dtype, geom):
    """bs_projector_level (round 6: one level of the bins head's projector path in one launch) against the launches it replaces --
    bs_resize_bias_relu_nhwc, the 3-pass pair bs_gemm of the projector's second convolution, bs_add_resized, and (last level) the 3-pass
    bs_gemm of the log-binomial embedding -- and against torch in fp64.  Eh and emb come out of the same MFMA sequence as bs_gemm's: equal
    bits are expected and asserted to 1e-6; x skips the pair rounding of emb, so it may differ from the old path by one step of the 16-bit
    format on a rounding boundary.  The last geometry is the bench's finest level (more slices than the chip has waves: the persistent loop)."""
    B, Hl, Wl, last = geom
    H, W, PM, E, NE = 2 * Hl, 2 * Wl, 64, 128, 80
    M = B * H * W
    z = rnd(B, Hl, Wl, 2 * PM, seed=31, dtype=dtype)
    prev = rnd(B, Hl, Wl, 2 * E, seed=32, dtype=dtype)
    z[..., PM:] *= 2.0 ** -11           # lo halves of a realistic magnitude
    prev[..., E:] *= 2.0 ** -11
    b1, bc2, be = rnd(PM, seed=33), rnd(E, seed=34), rnd(NE, seed=35)

    def pack3(wf):                       # fp32 [N, K] -> [W_hi | W_hi | W_lo] (ZoeDepthEngine._wn)
        hi = wf.to(dtype)
        lo = (wf - hi.float()).to(dtype)
        return torch.cat([hi, hi, lo], 1).contiguous()

    wc2_f, we_f = rnd(E, PM, seed=36) / PM ** 0.5, rnd(NE, PM, seed=37) / PM ** 0.5
    wc2, we = pack3(wc2_f), pack3(we_f)
    x = torch.full((M, E), 7.0, device=dev(), dtype=dtype)
    emb = None if last else torch.full((M, 2 * E), 7.0, device=dev(), dtype=dtype)
    eh = torch.full((M, NE), -7.0, device=dev()) if last else None
    L.projector_level(z, b1, prev, wc2, bc2, we if last else None, be if last else None, x, emb, eh, B, Hl, Wl, H, W, PM, E, NE)
    # ---- the launches it replaces
    e1 = torch.empty(M, 2 * PM, device=dev(), dtype=dtype)
    L.resize_bias_relu_nhwc(z, b1, e1, B, Hl, Wl, PM, H, W, split=True)
    emb32 = torch.empty(M, E, device=dev())
    L.gemm(e1, wc2, emb32, M=M, N=E, K=3 * PM, lda=2 * PM, seg1=PM, bias=bc2)
    emb_old = torch.empty(M, 2 * E, device=dev(), dtype=dtype)
    L.gemm(e1, wc2, emb_old, M=M, N=E, K=3 * PM, lda=2 * PM, seg1=PM, bias=bc2, ldo=2 * E, out_split_off=E)
    y_old = torch.empty_like(emb_old)
    L.add_resized(emb_old, prev, y_old, B, Hl, Wl, H, W, E, split=True)
    torch.cuda.synchronize()
    x_old = y_old[:, :E].float()
    # one step of the 16-bit format, plus what the old path's pair rounding of emb (22 / 16 significant bits) is worth where emb and the
    # resampled previous embedding cancel
    f16 = dtype == torch.float16
    ulp = x_old.abs() * (2.0 ** -10 if f16 else 2.0 ** -7) + emb32.abs() * (2.0 ** -21 if f16 else 2.0 ** -15) + 1e-7
    dx = (x.float() - x_old).abs()
    assert (dx <= ulp).all(), f"x differs from the four-launch path by more than one step: {(dx / ulp).max().item():.2f}"
    frac = (dx > 0).float().mean().item()
    assert frac < (5e-3 if f16 else 3e-2), f"{frac:.2e} of x's elements differ from the four-launch path"
    if emb is not None:
        assert torch.equal(emb, emb_old), f"emb pairs: max |diff| {(emb.float() - emb_old.float()).abs().max().item():.3e}"
    if last:
        eh_old = torch.empty(M, NE, device=dev())
        L.gemm(e1, we, eh_old, M=M, N=NE, K=3 * PM, lda=2 * PM, seg1=PM, bias=be)
        torch.cuda.synchronize()
        d = (eh - eh_old).abs().max().item()
        assert d <= 1e-6 * max(1.0, eh_old.abs().max().item()), f"Eh: max |diff| to the 3-pass bs_gemm {d:.3e}"
    # ---- fp64
    zf = (z[..., :PM].double() + z[..., PM:].double()).permute(0, 3, 1, 2)
    pf = (prev[..., :E].double() + prev[..., E:].double()).permute(0, 3, 1, 2)
    e64 = torch.relu(F.interpolate(zf, size=(H, W), mode="bilinear", align_corners=True) + b1.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).reshape(M, PM)
    emb64 = e64 @ wc2_f.double().t() + bc2.double()
    x64 = emb64 + F.interpolate(pf, size=(H, W), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).reshape(M, E)
    ex = ((x.double() - x64).abs() / (x64.abs() + 1.0)).max().item()
    assert ex < (2e-3 if dtype == torch.float16 else 1.6e-2), ex          # one 16-bit rounding of the sum
    e_emb = (emb32.double() - emb64).abs().max().item()
    assert e_emb < (5e-5 if dtype == torch.float16 else 5e-3), e_emb     # (the 3-pass pair product itself: ~22 / ~16 significant bits, the maximum over 1e7 values)
    if last:
        eh64 = e64 @ we_f.double().t() + be.double()
        ee = (eh.double() - eh64).abs().max().item()
        assert ee < (5e-5 if dtype == torch.float16 else 5e-3), ee
    report(f"projector_level {dtype} B={B} {Hl}x{Wl}->{H}x{W} last={last}: x differs from the four-launch path on {frac:.2e} of its elements (<= one step), "
           f"max rel err vs fp64 {ex:.2e}")
