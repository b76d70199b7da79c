#!/usr/bin/env python3
"""CPU study on the oracle (no GPU): how much of the depth error caused by rounding the BEiT weights to fp16 is the
token-independent (rank-1) part  1 * (mean_tokens(A)^T dW)?  Variants of every backbone Linear (q/k/v/o/fc1/fc2):
  exact      F.linear(x, W)
  rounded    F.linear(x, W16)                                  W16 = fp16(W)
  mean       F.linear(x, W16) + mean_t(x) @ dW^T               dW = W - W16, mean over the tokens of each image
  mean_p     as mean, but cls row excluded from the mean and corrected exactly (the cls tile keeps its FP8 correction pass)
Prints depth L1 / mean signed error vs exact, in metres.  Usage: python tools/probes/weight_mean_correction.py [seed ...]"""
import os
import sys
import time

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import zoedepth_ref as Z          # noqa: E402
from bodyslam_amd.synthetic import make_sequence   # noqa: E402


class FProxy:
    def __init__(self, mode, classes):
        self.mode, self.classes, self.cache = mode, classes, {}

    def __getattr__(self, name):
        return getattr(TF, name)

    def linear(self, x, W, b=None):
        if self.mode == "exact" or x.dim() != 3 or x.shape[1] < 700 or W.shape not in self.classes:
            return TF.linear(x, W, b)
        key = W.data_ptr()
        if key not in self.cache:
            W16 = W.half().float()
            self.cache[key] = (W16, W - W16)
        W16, dW = self.cache[key]
        y = TF.linear(x, W16, b)
        if self.mode == "rounded":
            return y
        if self.mode == "mean":
            return y + TF.linear(x.mean(dim=1, keepdim=True), dW)
        if self.mode == "mean_p":
            sub = int(os.environ.get("SUBSAMPLE", "1"))
            y = y + TF.linear(x[:, 1::sub].mean(dim=1, keepdim=True), dW)
            y[:, 0] = TF.linear(x[:, 0], W, b)
            return y
        if self.mode in ("mean_p_a16", "blk_p_a16", "blk_p"):
            # patch rows: fp16-rounded activations x fp16-rounded weights (one 16-bit pass) + the rank-1 / block-mean correction;
            # cls row exact (it keeps both FP8 correction passes)
            xp = x[:, 1:]
            xr = xp.half().float() if self.mode.endswith("a16") else xp
            yp = TF.linear(xr, W16, b)
            if self.mode.startswith("mean"):
                sub = int(os.environ.get("SUBSAMPLE", "1"))          # mean over every sub-th patch token only
                yp = yp + TF.linear(xp[:, ::sub].mean(dim=1, keepdim=True), dW)
            else:                                   # means over 6x8-patch blocks of the 24x32 grid (16 per image)
                B, T, K = xp.shape
                hp, wp = 24, T // 24
                g = xp.view(B, 4, hp // 4, 4, wp // 4, K).mean(dim=(2, 4))                      # [B, 4, 4, K]
                c = TF.linear(g, dW)                                                            # [B, 4, 4, N]
                c = c[:, :, None, :, None, :].expand(B, 4, hp // 4, 4, wp // 4, c.shape[-1]).reshape(B, T, -1)
                yp = yp + c
            return torch.cat([TF.linear(x[:, :1], W, b), yp], dim=1)
        raise ValueError(self.mode)


MODES = os.environ.get("MODES", "rounded,mean,mean_p,mean_p_a16,blk_p,blk_p_a16").split(",")


def main(seeds):
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = Z.ZoeConfig()
    H = cfg.hidden
    shapes = {"qkv": (H, H), "fc1": (cfg.intermediate, H), "fc2": (H, cfg.intermediate)}
    frames = torch.from_numpy(make_sequence(1, 480, 640, seed=3))
    for seed in seeds:
        w = Z.synth_weights(cfg, seed=seed)
        t0 = time.time()
        Z.F = FProxy("exact", set())
        ref = Z.infer_depth(w, cfg, frames)
        print(f"seed {seed}: exact forward {time.time() - t0:.1f} s, depth range {ref.min():.3f}..{ref.max():.3f}", flush=True)
        allc = {torch.Size(s) for s in shapes.values()}
        sets = {"all": allc}
        if os.environ.get("PER_CLASS"):
            sets = {k: {torch.Size(v)} for k, v in shapes.items()}          # ("qkv" = q, k, v and o: same shape)
        if os.environ.get("NO_FC1"):
            sets = {"qkv+o+fc2": {torch.Size(shapes["qkv"]), torch.Size(shapes["fc2"])}}
        for cname, cset in sets.items():
            for mode in MODES:
                Z.F = FProxy(mode, cset)
                d = Z.infer_depth(w, cfg, frames)
                e = d - ref
                print(f"seed {seed} {cname:4s} {mode:10s}: L1 {e.abs().mean():.3e} max {e.abs().max():.3e} signed {e.mean():+.3e}", flush=True)
        Z.F = TF


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [1, 2])
