"""MPEM on MI355X: the CyclePose generator's pose branch as a plan of HIP launches.

Reference: ConditionalGenerator.forward(mode="pose") BodySLAM_not_refactored/MPEM/architecture_v3.py:195-226
(initial_model :120-125, downsampling :129-139, pose_conv :143-147, pose_dense :150-155, skip_linear
:208-211) with the input transform of MPEMInterface.infer_relative_pose_between
(MPEM/mpem_interface.py:40-44,85-94).  Parameter names are the reference's state-dict names.

Quirk Q1: the reference creates ``skip_linear`` lazily inside forward, AFTER
load_state_dict(strict=False) ran, so a fresh MPEMInterface uses randomly initialised skip weights.
Here ``skip_linear.weight/bias`` are explicit weights (taken from the checkpoint when present).

Layout: 7x7 conv as a GEMM over a reflect-padded patch matrix [P*128*128, 320] (K = 294 padded);
the three stride-2 3x3 convs as implicit GEMM over NHWC; conv outputs fp32 (InstanceNorm statistics
are taken in fp32), normalised activations 16-bit -- (hi | lo) pairs in the default accurate mode, where
every convolution is a split-precision product; the 262 656-long skip dot products in fp32.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib as L

CROP = 128
K0 = 320
SKIP_FEATURES = 512 + 256 * 32 * 32


class CyclePoseEngine:
    def __init__(self, weights: Dict[str, torch.Tensor], dtype=torch.float16, device: int = 0, precision: str = "accurate"):
        """precision: "accurate" (default) -- the four convolutions as split-precision products: activations and weights are
        (hi, lo) pairs of 16-bit values (x = hi + lo to ~22 bits) and one bs_gemm launch evaluates A_hi W_hi + A_lo W_hi +
        A_hi W_lo (K segments; 2.43 GFLOP per pair, 0.16 % of the loop's work, so the three passes are free); relative pose
        within 1e-5 of the fp32 reference.  "fast": one 16-bit pass per product (~1e-3 at fp16)."""
        L.init(device)
        assert dtype in (torch.float16, torch.bfloat16) and precision in ("accurate", "fast")
        self.dtype = dtype
        self.acc = precision == "accurate"
        self.dev = torch.device("cuda", device)
        self._plans = {}
        g = lambda k: weights[k].detach().float()
        h = lambda t: t.to(self.dev, dtype=dtype).contiguous()
        f = lambda t: t.to(self.dev, dtype=torch.float32).contiguous()
        cw = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)          # [O,I,kh,kw] -> [O][(ky,kx,ci)]

        def split(t):
            hi = t.to(dtype)
            return hi, (t - hi.float()).to(dtype)

        w = self.w = {}
        w0 = cw(g("initial_model.1.weight"))                                   # [64, 294]
        w0 = torch.cat([w0, torch.zeros(w0.shape[0], K0 - w0.shape[1])], 1)
        if self.acc:                                                           # [W_hi | W_hi | W_lo] against A = [hi | lo], then hi again
            hi, lo = split(w0)
            w["c0.w"] = torch.cat([hi, hi, lo], 1).to(self.dev).contiguous()
        else:
            w["c0.w"] = h(w0)
        w["c0.b"] = f(g("initial_model.1.bias"))
        for name, key in (("c1", "downsampling.0"), ("c2", "downsampling.3"), ("c3", "pose_conv.0")):
            k = g(key + ".weight").permute(0, 2, 3, 1)                         # [O, kh, kw, I]
            if self.acc:    # segment 0 = [W_hi | W_hi] against the 2I (hi | lo) channels, segment 1 = W_lo against the hi channels
                hi, lo = split(k)
                w[name + ".w"] = torch.cat([L.conv_weight(torch.cat([hi, hi], -1)), L.conv_weight(lo)], 1).to(self.dev).contiguous()
            else:
                w[name + ".w"] = h(L.conv_weight(k))                           # conv mode K order: chunk, tap, channel
            w[name + ".b"] = f(g(key + ".bias"))
        if "skip_linear.weight" not in weights:
            raise KeyError("skip_linear.weight missing: the reference would silently use a random layer here "
                           "(architecture_v3.py:208-209); pass it explicitly")
        # skip_linear is sized by the network input (architecture_v3.py:205-209): one weight per input window, keyed by the
        # number of pixels of the stride-4 map (32*32 for the 128x128 crop)
        self._skip, self._skip_shape = {}, {}
        n_skip = (g("skip_linear.weight").shape[1] - 512) // 256
        # (the checkpoint's own layer belongs to the reference's 128 x 128 crop: a 32 x 32 map)
        self.add_skip(g("skip_linear.weight"), g("skip_linear.bias"), map_hw=(32, 32) if n_skip == 1024 else None)
        w["d1.w"], w["d1.b"] = f(g("pose_dense.1.weight")), f(g("pose_dense.1.bias"))
        w["d2.w"], w["d2.b"] = f(g("pose_dense.3.weight")), f(g("pose_dense.3.bias"))

    # (keyed by the number of map positions h' * w', as the reference's lazily created Linear(512 + 256 h' w', 7) is: the flattened NCHW
    # feature index c * h'w' + p only knows the row-major position p, which is the row of the [7][h'w'][C] layout for every (h', w')
    # of that product)
    def add_skip(self, weight: torch.Tensor, bias: torch.Tensor, map_hw=None) -> int:
        """Register a skip_linear weight [7, 512 + 256*h*w] (h x w = the stride-4 map of the network input); returns h*w.
        map_hw = (h, w) the weight was trained for, when known: a plan whose map has the same number of positions but another shape
        (32 x 43 against 43 x 32) then warns -- the arithmetic is what the reference's Linear would do with it (it only sees the
        flattened index), the result is meaningless."""
        ws = weight.detach().float()
        assert ws.dim() == 2 and ws.shape[0] == 7 and (ws.shape[1] - 512) % 256 == 0 and ws.shape[1] > 512, ws.shape
        hw = (ws.shape[1] - 512) // 256
        assert map_hw is None or map_hw[0] * map_hw[1] == hw, (map_hw, hw)
        f = lambda t: t.to(self.dev, dtype=torch.float32).contiguous()
        self._skip[hw] = (f(ws[:, :512]), f(ws[:, 512:].view(7, 256, hw).permute(0, 2, 1)),      # NCHW flatten -> [7][HW][C]
                          f(bias.detach().float()))
        self._skip_shape[hw] = tuple(map_hw) if map_hw is not None else None
        return hw

    @staticmethod
    def map_hw(ch: int, cw: int):
        """spatial size after the two stride-2 convolutions (k3 p1): the map skip_linear flattens"""
        h1, w1 = (ch - 1) // 2 + 1, (cw - 1) // 2 + 1
        return (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1

    def plan_for(self, N: int, P: int, H: int, W: int, window=None) -> "_PosePlan":
        key = (N, P, H, W, window)
        if key not in self._plans:
            self._plans[key] = _PosePlan(self, N, P, H, W, window)
        return self._plans[key]

    def infer_pairs(self, frames_u8: torch.Tensor, pairs: torch.Tensor, taps: Optional[dict] = None, window=None) -> torch.Tensor:
        """frames uint8 [N,H,W,3] (GPU), pairs int32 [P,2] (indices into frames: prev, curr)
        -> relative poses fp32 [P,4,4] (the plan's static buffer).  window: None = the reference's CenterCrop(128); "full" = the
        whole frame is the network input (type_of_trans='resize': frames already resized to 128 x W'), or (top, left, h, w)."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and pairs.dtype == torch.int32
        N, H, W, _ = frames_u8.shape
        P = pairs.shape[0]
        if P == 0:      # no pair: nothing to launch
            return torch.empty(0, 4, 4, device=self.dev)
        if window == "full":
            window = (0, 0, H, W)
        plan = self.plan_for(N, P, H, W, window)
        plan.frames.copy_(frames_u8)
        plan.pairs.copy_(pairs)
        plan.plan.run(taps)
        return plan.T.view(P, 4, 4)


class _PosePlan:
    def __init__(self, eng: CyclePoseEngine, N: int, P: int, H: int, W: int, window=None):
        w, dt_, dev = eng.w, eng.dtype, eng.dev
        CH, CW = (CROP, CROP) if window is None else (window[2], window[3])
        h2, w2 = eng.map_hw(CH, CW)
        if h2 * w2 not in eng._skip:
            raise KeyError(f"no skip_linear weight for a {CH}x{CW} network input ({512 + 256 * h2 * w2} features): "
                           "register one with CyclePoseEngine.add_skip (the reference would create a random layer, architecture_v3.py:208-209)")
        if eng._skip_shape.get(h2 * w2) not in (None, (h2, w2)):
            import warnings
            warnings.warn(f"CyclePose: the skip_linear weight of {h2 * w2} map positions was registered for a {eng._skip_shape[h2 * w2]} map and is "
                          f"being applied to a {(h2, w2)} one: same flattened size, other geometry (the reference's Linear would do the same)")
        skip_pool, skip_x2, skip_b = eng._skip[h2 * w2]
        H1, W1 = (CH - 1) // 2 + 1, (CW - 1) // 2 + 1
        H3, W3 = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
        e16 = lambda *s: torch.empty(*s, device=dev, dtype=dt_)
        e32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        self.frames = torch.empty(N, H, W, 3, device=dev, dtype=torch.uint8)
        self.pairs = torch.zeros(P, 2, device=dev, dtype=torch.int32)
        Pl = self.plan = L.Plan(dev)
        in_scratch = e32(P * ((CH * CW + 255) // 256) * 2 * 256)
        acc = eng.acc
        m2, SP = (2, 16) if acc else (1, 0)           # pair multiplier / the producers' "split" flag

        def conv(name, A, out, hin, win, ci, co, **kw):
            ho, wo = (hin - 1) // 2 + 1, (win - 1) // 2 + 1
            if acc:
                Pl.gemm(name, A, w[name + ".w"], out, M=P * ho * wo, N=co, K=9 * ci * 3, lda=2 * ci, seg1=ci,
                        conv=(hin, win, 2 * ci, ho, wo, 3, 3, 2, 1, 1), bias=w[name + ".b"],
                        precision_passes=3, **kw)
            else:
                Pl.gemm(name, A, w[name + ".w"], out, M=P * ho * wo, N=co, K=9 * ci, lda=ci,
                        conv=(hin, win, ci, ho, wo, 3, 3, 2, 1, 1), bias=w[name + ".b"], **kw)

        cols = e16(P * CH * CW, K0 * m2)
        if window is None:
            Pl.add("im2col", "bs_cyclepose_im2col", self.frames, self.pairs, cols, P, H, W, L.dt(cols) | SP)
        else:
            Pl.add("im2col", "bs_cyclepose_im2col_window", self.frames, self.pairs, cols, P, H, W, window[0], window[1], CH, CW, L.dt(cols) | SP)
        c0 = e32(P, CH, CW, 64)
        Pl.gemm("c0", cols, w["c0.w"], c0, M=P * CH * CW, N=64, K=K0 * (3 if acc else 1), lda=K0 * m2, seg1=K0 if acc else 0,
                bias=w["c0.b"], precision_passes=3 if acc else 1)
        a0 = e16(P, CH, CW, 64 * m2)
        Pl.add("in0", "bs_instnorm_relu_nhwc", c0, a0, None, in_scratch, P, CH * CW, 64, 1e-5, L.dt(a0) | SP)
        Pl.mark("c0", a0, ("nhwc", P, CH, CW, 64, 1 if acc else 0))
        c1 = e32(P, H1, W1, 128)
        conv("c1", a0, c1, CH, CW, 64, 128)
        a1 = e16(P, H1, W1, 128 * m2)
        Pl.add("in1", "bs_instnorm_relu_nhwc", c1, a1, None, in_scratch, P, H1 * W1, 128, 1e-5, L.dt(a1) | SP)
        c2 = e32(P, h2, w2, 256)
        conv("c2", a1, c2, H1, W1, 128, 256)
        a2 = e16(P, h2, w2, 256 * m2)
        x2 = e32(P, h2, w2, 256)
        Pl.add("in2", "bs_instnorm_relu_nhwc", c2, a2, x2, in_scratch, P, h2 * w2, 256, 1e-5, L.dt(a2) | SP)
        Pl.mark("c2", x2, ("nhwc", P, h2, w2, 256))
        c3 = e32(P, H3, W3, 512)
        conv("c3", a2, c3, h2, w2, 256, 512, act=L.ACT_RELU)
        pooled = e32(P, 512)
        Pl.add("pool", "bs_avgpool_nhwc", c3, pooled, P, H3 * W3, 512)
        Pl.mark("pooled", pooled, ("raw",))
        self.pose7 = e32(P, 7)
        self.T = e32(P, 16)
        scratch = e32(P * ((h2 * w2 * 256 + 4095) // 4096) * 8)
        Pl.add("head", "bs_cyclepose_head", pooled, x2, skip_pool, skip_x2, skip_b, w["d1.w"], w["d1.b"], w["d2.w"], w["d2.b"],
               self.pose7, self.T, scratch, P, h2 * w2, 256)
