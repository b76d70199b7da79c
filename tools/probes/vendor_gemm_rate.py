"""Probe: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the backbone's GEMM shapes, fp16 in / fp16 out, no
epilogue -- a reference point for bs_gemm's main loop (tools/bench_kernels.py tile 409 = bs_gemm without its epilogue).  Not a product path."""
import torch
dev = torch.device("cuda:0")
M = 256 + 128 * 768
for name, N, K in (("qkv", 3072, 1024), ("o", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("conv-as-gemm 256->256 3x3", 256, 2304)):
    Mx = M if not name.startswith("conv") else 128 * 192 * 256
    a = torch.randn(Mx, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    print(f"{name:28s} M={Mx} N={N} K={K}: {us:9.1f} us  {2.0 * Mx * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)
