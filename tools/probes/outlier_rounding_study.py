#!/usr/bin/env python3
"""CPU study on the oracle (no GPU): where does accurate mode's depth error on OUTLIER-CHANNEL weights come from?
(tests/test_zoedepth_gpu.py::_hook_outlier_channels: 6 channels 50x larger after every LayerNorm, damped again in q / fc1.)
The all-"full" product still rounds, once each, to 16 bits: Q, K, V, the softmax probabilities P (un-normalised exp2, fp16) -- and
carries every GEMM operand as hi16 + an e4m3 correction.  Each variant below applies ONE of those roundings inside the fp32 oracle
and prints the depth L1 against the exact oracle, in metres:
  q16 / k16 / v16 / p16       the operand rounded to fp16 (what attention_tab2_kernel consumes)
  qk16, qkvp16                combinations
  qcorr, kcorr, vcorr, pcorr  the operand as hi16 + fp16(residual), unscaled (the corrected attention's operand pair)
  act_f8                      every backbone GEMM's A operand as hi16 + e4m3((x - hi16) * 2^11)   (the "full" product's A side)
  pe_f8 / pe_16               the patch embedding's operands as hi16 + e4m3 residuals / as single 16-bit values
  mh16                        the bins head's single-pass 1x1 convs (conv2, seed regressors, seed projector, attractor MLPs) on 16-bit operands
  w_f8                        every backbone weight as W16 + e4m3((W - W16) * 2^b), b per matrix   (the "full" product's W side)
Usage: python tools/probes/outlier_rounding_study.py [variant ...]      (about 20 s per variant on 8 cores)"""
import os
import sys
import time

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import zoedepth_ref as Z                  # noqa: E402
from bodyslam_amd.synthetic import make_sequence      # noqa: E402
import test_zoedepth_gpu as T                         # noqa: E402

MH_PTRS = set()
HOOKS = {"outlier": T._hook_outlier_channels, "layerscale": T._hook_layerscale_wide, "heavy": T._hook_heavy_tailed, "none": None}


def r16(x):
    return x.half().float()


def e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def hi_lo8(x, lo_exp=11):
    h = r16(x)
    return h + e4m3((x - h) * 2.0 ** lo_exp) * 2.0 ** -lo_exp


def hi_lo16(x):
    """hi16 + an UNSCALED fp16 residual (subnormals kept): the corrected attention's operand pair"""
    h = r16(x)
    return h + r16(x - h)


class TorchProxy:
    """stands in for the `torch` module inside oracle.zoedepth_ref: matmul / softmax see the attention operands"""

    def __init__(self, mode):
        self.mode = mode
        self.stage = 0

    def __getattr__(self, name):
        return getattr(torch, name)

    def matmul(self, a, b):
        m = self.mode
        if a.dim() == 4 and a.shape[-1] == 64 and b.shape[-2] == 64 and a.shape[2] > 700:      # q @ k^T
            if "q16" in m or "qk16" in m or "qkvp16" in m:
                a = r16(a)
            if "k16" in m or "qk16" in m or "qkvp16" in m:
                b = r16(b)
            if "kcorr" in m:
                b = hi_lo16(b)
            if "qcorr" in m:
                a = hi_lo16(a)
            return torch.matmul(a, b)
        if a.dim() == 4 and a.shape[-1] == a.shape[-2] and a.shape[2] > 700:                   # p @ v
            if "p16" in m or "qkvp16" in m:
                # the kernel rounds the un-normalised exp2(s - max): same relative rounding as rounding p itself
                a = r16(a)
            if "v16" in m or "qkvp16" in m:
                b = r16(b)
            if "vcorr" in m:
                b = hi_lo16(b)
            if "pcorr" in m:
                a = hi_lo16(a)
            return torch.matmul(a, b)
        return torch.matmul(a, b)


class FProxy:
    def __init__(self, mode):
        self.mode, self.cache = mode, {}

    def __getattr__(self, name):
        return getattr(TF, name)

    def conv2d(self, x, W, b=None, **kw):
        m = self.mode
        if "mh16" in m and W.data_ptr() in MH_PTRS:               # the bins head's single-precision 1x1 convs (16-bit operands)
            return TF.conv2d(r16(x), r16(W), b, **kw)
        if W.shape[-1] == 16 and kw.get("stride") == 16:           # the patch embedding
            if "pe_f8" in m:        # A = hi16 + e4m3 residual, W = W16 + e4m3 residual (the "full" product's operands)
                W16 = r16(W)
                d = W - W16
                bexp = torch.floor(torch.log2(448.0 / d.abs().max().clamp_min(1e-30)))
                return TF.conv2d(hi_lo8(x), W16 + e4m3(d * 2.0 ** bexp) * 2.0 ** -bexp, b, **kw)
            if "pe_16" in m:
                return TF.conv2d(r16(x), r16(W), b, **kw)
        return TF.conv2d(x, W, b, **kw)

    def linear(self, x, W, b=None):
        m = self.mode
        big = x.dim() == 3 and x.shape[1] > 700 and W.shape[0] >= 1024 and W.shape[1] >= 1024
        if not big:
            return TF.linear(x, W, b)
        if "act_f8" in m:
            x = hi_lo8(x)
        if "act16" in m:
            x = r16(x)
        if "w_f8" in m:
            key = W.data_ptr()
            if key not in self.cache:
                W16 = r16(W)
                d = W - W16
                bexp = torch.floor(torch.log2(448.0 / d.abs().max().clamp_min(1e-30)))
                self.cache[key] = W16 + e4m3(d * 2.0 ** bexp) * 2.0 ** -bexp
            W = self.cache[key]
        if "w16" in m:
            key = W.data_ptr()
            if key not in self.cache:
                self.cache[key] = r16(W)
            W = self.cache[key]
        return TF.linear(x, W, b)


def main(argv):
    hook_name = os.environ.get("HOOK", "outlier")
    seed = int(os.environ.get("SEED", "9"))
    variants = argv or ["q16", "k16", "v16", "p16", "qk16", "qkvp16", "act_f8", "w_f8", "qkvp16+act_f8+w_f8"]
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = Z.ZOED_NK
    w = Z.synth_weights(cfg, seed=seed)
    if HOOKS[hook_name] is not None:
        HOOKS[hook_name](w)
    frames = torch.from_numpy(make_sequence(1, 480, 640, seed=seed))
    # single 16-bit GEMMs of the bins head in every engine mode: conv2, seed regressors / projector conv1+conv2, attractor MLPs
    sel = os.environ.get("MH_SEL", "")             # substring filter on the weight key (e.g. "attractors", "seed_bin", "conv2")
    for k_, v_ in w.items():
        if k_.startswith("metric_head.") and v_.dim() == 4 and ("projectors" not in k_) and ("conditional_log_binomial" not in k_) and sel in k_:
            MH_PTRS.add(v_.data_ptr())
    print("mh16 applies to", len(MH_PTRS), "weights", flush=True)
    out = open(os.path.join(ROOT, "gpurun_out", "outlier_rounding_study.txt"), "a")
    with torch.no_grad():
        t0 = time.time()
        ref = Z.infer_depth(w, cfg, frames)
        print(f"hook {hook_name} seed {seed}: exact forward {time.time() - t0:.1f} s, depth {ref.min():.3f}..{ref.max():.3f}", flush=True)
        for v in variants:
            Z.torch = TorchProxy(v)
            Z.F = FProxy(v)
            t0 = time.time()
            d = Z.infer_depth(w, cfg, frames)
            e = d - ref
            line = f"hook {hook_name} seed {seed} {v:28s}: L1 {e.abs().mean():.3e} max {e.abs().max():.3e} signed {e.mean():+.3e}  ({time.time() - t0:.0f} s)"
            print(line, flush=True)
            out.write(line + "\n")
            out.flush()
    Z.torch, Z.F = torch, TF


if __name__ == "__main__":
    main(sys.argv[1:])
