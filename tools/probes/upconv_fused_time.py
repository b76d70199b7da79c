"""Probe: the relative head's upsample + conv2 at the bench shape (NB = 128 images, 192 x 256 -> 384 x 512, 128 -> 32 channels): bs_upconv_fused
against the two launches it replaces (bs_gemm tap products on the 128 x 64 tile + bs_upconv_tapsum).   python tools/probes/upconv_fused_time.py [NB]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bodyslam_amd import _lib as L

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W, C, Co = 192, 256, 128, 32
L.init(0)
dev = torch.device("cuda:0")
dt = torch.float16
M = NB * H * W
x32 = torch.randn(M // 8, C, device=dev)
xin = torch.empty(M, 2 * C, device=dev, dtype=dt)
for i in range(8):
    L.cast_split(x32, xin[i * (M // 8):(i + 1) * (M // 8)], M // 8, C, f8=True)
del x32
w2 = torch.randn(9 * Co, C) * 0.05
wp, (sb0, sb1) = L.f8_weight(w2, dt)
wp = wp.to(dev)
bias = torch.randn(Co, device=dev)
scales = (127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1)
out = torch.zeros(NB, 2 * H, 2 * W, 2 * Co, device=dev, dtype=dt)
y9 = torch.empty(M, 9 * Co, device=dev, dtype=torch.float32)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if os.environ.get("ABLATE"):       # diagnostics: bits of `relu` beyond bit 0 switch parts of the kernel off (csrc/upconv_fused.hip)
    for ab in (0, 2, 32, 2 | 32, 4, 8, 16, 4 | 8, 2 | 4 | 8 | 32, 2 | 4 | 8 | 16 | 32):
        t = timed(lambda: L.upconv_fused(xin.view(NB, H, W, 2 * C), wp, bias, out, NB, H, W, C, Co, mode=1, split=2, relu=1 | ab, f8_scales=scales))
        print(f"ablate {ab:3d} (2 no weight re-DMA, 32 no window re-DMA, 4 no interpolation, 8 no MFMA, 16 no stores): {t:.3f} ms", flush=True)
    sys.exit(0)
for mode in (1, 2):
    t_f = timed(lambda: L.upconv_fused(xin.view(NB, H, W, 2 * C), wp, bias, out, NB, H, W, C, Co, mode=mode, split=2, relu=True, f8_scales=scales))
    a = out.clone()
    t_g = timed(lambda: L.gemm(xin, wp, y9, M=M, N=9 * Co, K=C, lda=2 * C, f8_seg=2 * C, f8_wonly_from=-1 if mode == 1 else 0, f8_scales=scales, tile=2))
    t_t = timed(lambda: L.upconv_tapsum(y9, bias, out, NB, H, W, Co, 2 * H, 2 * W, True, 2, True))
    d = (a[..., :Co].float() - out[..., :Co].float()).abs().max().item()
    gb = (M * 4 * C + NB * 4 * H * W * 4 * Co) / 1e9
    print(f"NB={NB} mode {mode}: fused {t_f:.3f} ms ({gb / t_f:.2f} TB/s of its {gb:.1f} GB in + out) | gemm {t_g:.3f} + tapsum {t_t:.3f} = {t_g + t_t:.3f} ms"
          f" | max |hi16 difference| {d:.2e}", flush=True)
