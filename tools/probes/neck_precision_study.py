#!/usr/bin/env python3
"""CPU study on the oracle (no GPU): which correction products do the 3x3 convolutions of the DPT neck / relative head need so
that the depth map stays within 1e-4 m of the fp32 result, and how few bits may the correction operands have?

The backbone runs exact (its hidden states are computed once per seed); every 3x3 Conv2d of the neck, the fusion stage and the
relative head is replaced by one of:
  single      conv(a16, w16)                                  a16 = fp16(a), w16 = fp16(w): one 16-bit pass
  full8       + conv(q8(a16), q8(dw)) + conv(q8(da), q8(w16))  e4m3 operands with one power-of-two scale per tensor (the round-2 product)
  a8          single + the activation-rounding correction only
  w8          single + the weight-rounding correction only
  a8_wmean    a8 + the weight-rounding term of the per-image mean activation (conv of the constant mean image with dw, borders exact)
  a8_wmeanI   as a8_wmean, the mean term as a per-image bias (interior value used at the borders too)
  a8_wlocK / a8_wstripK   a8 + the weight-rounding term of the activation map averaged over K x K blocks / 1 x K strips (piecewise constant)
  mx6 / mx4   both corrections with e2m3 / e2m1 operands and one E8M0 scale per 32 channels (OCP MX blocks)
  amx6_wmean, amx4_wmean, mx4a_8w ...  mixtures (see MODES)
Prints depth L1 / max / mean signed error vs exact, in metres.
Usage: python tools/probes/neck_precision_study.py [seed ...]      (env MODES=comma list)"""
import os
import sys
import time

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import zoedepth_ref as Z          # noqa: E402
from bodyslam_amd.synthetic import make_sequence   # noqa: E402


def r16(x):
    return x.half().float()


def q8_tensor(x):
    """e4m3 with one power-of-two scale for the whole tensor (amax -> [128, 256))"""
    amax = x.abs().max().clamp_min(1e-30)
    s = 2.0 ** (7 - torch.floor(torch.log2(amax)))
    return (x * s).clamp(-448, 448).to(torch.float8_e4m3fn).float() / s


def minifloat(x, mbits, emax=2):
    """round to a sign + 2-bit exponent + mbits mantissa value (bias 1: normals 1 .. (2 - 2^-mbits) * 2^emax, subnormal step 2^-mbits)"""
    ax = x.abs()
    e = torch.floor(torch.log2(ax.clamp_min(1e-30))).clamp(0, emax)
    step = 2.0 ** (e - mbits)
    q = torch.round(ax / step) * step
    return torch.sign(x) * q.clamp_max((2.0 - 2.0 ** -mbits) * 2.0 ** emax)


def mx(x, mbits, dim, block=32):
    """OCP MX quantisation along `dim`: one E8M0 scale per `block` elements, elements e2m{mbits}"""
    x = x.movedim(dim, -1)
    sh = x.shape
    xb = x.reshape(*sh[:-1], sh[-1] // block, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = 2.0 ** (torch.floor(torch.log2(amax)) - 2)
    q = minifloat(xb / s, mbits) * s
    return q.reshape(sh).movedim(-1, dim)


class Proxy:
    def __init__(self, mode):
        self.mode = mode
        self.cache = {}

    def __getattr__(self, name):
        return getattr(TF, name)

    def _w(self, W):
        k = W.data_ptr()
        if k not in self.cache:
            W16 = r16(W)
            self.cache[k] = (W16, W - W16)
        return self.cache[k]

    def conv2d(self, x, W, b=None, stride=1, padding=0, **kw):
        m = self.mode
        if m == "exact" or W.shape[-1] != 3 or W.shape[1] < 64:
            return TF.conv2d(x, W, b, stride=stride, padding=padding, **kw)
        W16, dW = self._w(W)
        a16 = r16(x)
        da = x - a16
        cv = lambda a_, w_: TF.conv2d(a_, w_, None, stride=stride, padding=padding)
        y = TF.conv2d(a16, W16, b, stride=stride, padding=padding)
        if m == "single":
            return y
        parts = m.split("_")
        for p in parts:
            if p == "full8":
                y = y + cv(q8_tensor(a16), q8_tensor(dW)) + cv(q8_tensor(da), q8_tensor(W16))
            elif p == "a8":
                y = y + cv(q8_tensor(da), q8_tensor(W16))
            elif p == "w8":
                y = y + cv(q8_tensor(a16), q8_tensor(dW))
            elif p in ("mx6", "mx4", "mx5"):
                mb = {"mx6": 3, "mx5": 2, "mx4": 1}[p]
                y = y + cv(mx(a16, mb, 1, BLK), mx(dW, mb, 1, BLK)) + cv(mx(da, mb, 1, BLK), mx(W16, mb, 1, BLK))
            elif p == "mx4c":        # both corrections e2m1; the weight-rounding term splits into the exact per-image mean part and the
                                     # e2m1 product of the CENTRED activations
                abar = a16.mean(dim=(2, 3), keepdim=True)
                y = y + cv(abar.expand_as(a16).contiguous(), dW) + cv(mx(a16 - abar, 1, 1, BLK), mx(dW, 1, 1, BLK)) + cv(mx(da, 1, 1, BLK), mx(W16, 1, 1, BLK))
            elif p in ("amx6", "amx4", "amx5"):
                mb = {"amx6": 3, "amx5": 2, "amx4": 1}[p]
                y = y + cv(mx(da, mb, 1), mx(W16, mb, 1))
            elif p in ("wmx6", "wmx4", "wmx5"):
                mb = {"wmx6": 3, "wmx5": 2, "wmx4": 1}[p]
                y = y + cv(mx(a16, mb, 1), mx(dW, mb, 1))
            elif p == "wmean":       # conv of the per-image constant mean image with dW (zero padding: exact at the borders)
                mean_img = a16.mean(dim=(2, 3), keepdim=True).expand_as(a16)
                y = y + cv(mean_img, dW)
            elif p == "wmeanI":      # interior value as a per-image bias everywhere
                abar = a16.mean(dim=(2, 3))                                   # [B, C]
                y = y + torch.einsum("bc,oc->bo", abar, dW.sum(dim=(2, 3)))[:, :, None, None]
            elif p == "wmeanB":      # means over 4x4 blocks of the map (bilinear-free piecewise constant), borders exact
                B, C, H, Wd = a16.shape
                g = a16.view(B, C, 4, H // 4, 4, Wd // 4).mean(dim=(3, 5), keepdim=True).expand(B, C, 4, H // 4, 4, Wd // 4).reshape(B, C, H, Wd)
                y = y + cv(g, dW)
            elif p.startswith("wloc") or p.startswith("wstrip"):
                # the weight-rounding term of a LOCALLY averaged activation map: K x K blocks ("wlocK") or 1 x K strips ("wstripK"),
                # piecewise constant; what is left out is conv(a16 - local mean, dW) -- small where the feature maps are smooth
                strip = p.startswith("wstrip")
                K_ = int(p[6:] if strip else p[4:])
                B, C, H, Wd = a16.shape
                kh, kw_ = (1, K_) if strip else (K_, K_)
                ph, pw = (-H) % kh, (-Wd) % kw_
                ap = TF.pad(a16, (0, pw, 0, ph), mode="replicate")
                g = TF.avg_pool2d(ap, (kh, kw_))
                g = g.repeat_interleave(kh, 2).repeat_interleave(kw_, 3)[:, :, :H, :Wd]
                y = y + cv(r16(g), dW)
            else:
                raise ValueError(p)
        return y


BLK = int(os.environ.get("BLK", "32"))
RESQ = os.environ.get("RESQ", "")
MODES = os.environ.get("MODES", "single,full8,a8,w8,a8_wmean,a8_wmeanI,mx6,mx4,amx6_wmean,amx4_wmean,amx4_w8,a8_wmx4").split(",")


_orig_res_unit = Z._res_unit


def _res_unit_q(w, p, x):
    """the skip input as the consumer would reconstruct it from (hi16 | lo4): hi16 + e2m1 block-scaled residual"""
    y = Z.F.conv2d(Z.F.relu(x), w[p + "convolution1.weight"], w[p + "convolution1.bias"], padding=1)
    y = Z.F.conv2d(Z.F.relu(y), w[p + "convolution2.weight"], w[p + "convolution2.bias"], padding=1)
    x16 = r16(x)
    return y + x16 + mx(x - x16, 1, 1, BLK)


def run_tail(w, cfg, hiddens, hp, wp):
    fused, bott = Z.neck_forward(w, cfg, hiddens, hp, wp)
    rel, last = Z.relative_head_forward(w, fused[-1])
    xb = Z._c1(w, "metric_head.conv2", bott)
    logits = Z.router_logits(w, cfg, xb)
    route = torch.argmax(logits, dim=-1)
    B = xb.shape[0]
    out = torch.empty(B, last.shape[2], last.shape[3])
    for r, name in enumerate(cfg.head_names):
        sel = (route == r).nonzero().flatten()
        if sel.numel():
            out[sel] = Z.metric_head_single(w, cfg, name, xb[sel], [f[sel] for f in fused], last[sel])
    return out, route


def main(seeds):
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = Z.ZoeConfig()
    for seed in seeds:
        w = Z.synth_weights(cfg, seed=seed)
        frames = torch.from_numpy(make_sequence(1, 480, 640, seed=seed))
        with torch.no_grad():
            x = Z.preprocess(frames, (384, 512))
            x2 = torch.cat([x, torch.flip(x, dims=[3])])
            hp, wp = x.shape[2] // cfg.patch, x.shape[3] // cfg.patch
            t0 = time.time()
            Z.F = TF
            hiddens = Z.beit_forward(w, cfg, x2, None)
            print(f"seed {seed}: backbone {time.time() - t0:.1f} s", flush=True)

            def depth(mode):
                Z.F = Proxy(mode)
                Z._res_unit = _res_unit_q if (RESQ and mode != "exact") else _orig_res_unit
                out, route = run_tail(w, cfg, hiddens, hp, wp)
                Z._res_unit = _orig_res_unit
                Z.F = TF
                return Z.postprocess(out[:1], out[1:], 480, 640), route

            ref, route = depth("exact")
            print(f"seed {seed}: depth range {ref.min():.3f}..{ref.max():.3f} route {route.tolist()}", flush=True)
            for mode in MODES:
                t0 = time.time()
                d, _ = depth(mode)
                e = d - ref
                print(f"seed {seed} {mode:14s}: L1 {e.abs().mean():.3e} max {e.abs().max():.3e} signed {e.mean():+.3e}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [1, 5, 8])
