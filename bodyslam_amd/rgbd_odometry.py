"""Dense RGB-D odometry of the VO step: what ``VO._compute_vo_o3d`` gets from Open3D's
``rgbd_odometry_multi_scale(curr, prev, intrinsic, init, 1000.0, depth_max, [20, 10, 5], Method.Hybrid)``
(BodySLAM_not_refactored/3DM/visual_odometry.py:97-120): the relative pose between two RGB-D frames.

Open3D is not vendored and not installable offline; the algorithm is restated in oracle/rgbd_odometry_ref.py from the published
structure of its tensor hybrid odometry (parity with Open3D unpinned): intensity = grey / 255, depth = raw / depth_scale with values
<= 0 or > depth_max invalid, a 3-level pyramid ([1 4 6 4 1]^2 / 256; depth: the weights over the neighbours within 2 * 0.07 m of the
centre), target Sobel gradients / 8, coarse to fine with 20 / 10 / 5 Gauss-Newton iterations on the photometric + geometric
residuals, 0.07 m depth outlier gate, Huber deltas 0.1 / 0.05.

``association="nearest"`` (the default, the faithful mode): the target images and gradients are read at the NEAREST pixel of the
projected source point, and the robust step has Open3D's form (J^T J unweighted, J^T applied to the Huber-clipped residual).
``association="bilinear"`` (round 2's variant, kept as an option): the target is sampled bilinearly and the step is IRLS-weighted --
a smooth cost, which on sub-pixel motions converges to the micrometre where the nearest-pixel cost is piecewise constant.

Everything runs in the kernels of csrc/odometry.hip.  Three ways to call it:
  * ``RGBDOdometry()(curr_rgbd, prev_rgbd)`` -- what ``_compute_vo_o3d`` returns (the inverse of source -> target), one pair at a time;
  * ``track(color, depth)`` -- a stream of consecutive frames: each frame's pyramid and gradients are built ONCE (it is the source of
    one pair and the target of the next), the 11 + 70 launches of a frame are captured in a HIP graph and replayed, and the pose
    stays on the device, so a sequence is tracked without a host round trip per pair;
  * ``track_block(colors, depths)`` -- a block of consecutive frames at once.  The pairs of a sequence do not depend on each other
    (the reference starts every pair from the identity, visual_odometry.py:100), so n frames are n simultaneous pairs: every stage is
    ONE launch over the whole block (~100 launches per block instead of 81 per frame) and the work is HBM streaming instead of a
    chain of 10-microsecond launches.  Pair for pair the same sums in the same order as the other two: bit-equal results."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L

DEPTH_OUTLIER_TRUNC, DEPTH_HUBER, INTENSITY_HUBER = 0.07, 0.05, 0.1      # o3d.t.pipelines.odometry.OdometryLossParams defaults
ITERATIONS = (20, 10, 5)                                                   # visual_odometry.py:103-107, coarse -> fine
FLAG_NEAREST, FLAG_O3D_LOSS = 1, 2


def se3_exp(delta: np.ndarray) -> np.ndarray:
    w, v = np.asarray(delta[:3], dtype=np.float64), np.asarray(delta[3:], dtype=np.float64)
    th = float(np.linalg.norm(w))
    Wx = np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        R, V = np.eye(3) + Wx, np.eye(3) + 0.5 * Wx
    else:
        a, b, c = np.sin(th) / th, (1.0 - np.cos(th)) / th ** 2, (th - np.sin(th)) / th ** 3
        R = np.eye(3) + a * Wx + b * (Wx @ Wx)
        V = np.eye(3) + b * Wx + c * (Wx @ Wx)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, V @ v
    return T


class _Level:
    __slots__ = ("H", "W", "K", "Kc", "I", "D", "gIx", "gIy", "gDx", "gDy")     # maps: [H, W], or [slots, H, W] in track_block


class RGBDOdometry:
    def __init__(self, K: Sequence[float], device: int = 0, iterations: Sequence[int] = ITERATIONS, association: str = "nearest",
                 loss: Optional[str] = None):
        """K = (fx, fy, cx, cy) of the full-resolution frames.  association: "nearest" (Open3D's, default) | "bilinear";
        loss: "o3d" | "irls" (default: "o3d" with nearest, "irls" with bilinear)"""
        assert association in ("nearest", "bilinear")
        loss = loss or ("o3d" if association == "nearest" else "irls")
        assert loss in ("o3d", "irls")
        self.K = tuple(float(v) for v in K)
        self.dev = torch.device("cuda", device)
        self.iterations = tuple(int(i) for i in iterations)
        self.association, self.loss = association, loss
        self.flags = (FLAG_NEAREST if association == "nearest" else 0) | (FLAG_O3D_LOSS if loss == "o3d" else 0)
        L.init(device)
        self.last_trace = None
        self._trk = None
        self._blk = None

    # ---- device-side image preparation ---------------------------------------------------------------
    def _alloc_levels(self, H: int, W: int) -> List[_Level]:
        """the six maps of every pyramid level as views of ONE flat buffer (a frame's whole state moves with one copy)"""
        shapes = []
        h, w, k = H, W, self.K
        for _ in range(len(self.iterations)):
            shapes.append((h, w, k))
            h, w, k = (h + 1) // 2, (w + 1) // 2, tuple(v / 2.0 for v in k)
        flat = torch.empty(sum(6 * a * b for a, b, _ in shapes), device=self.dev, dtype=torch.float32)
        levels, off = [], 0
        for (h, w, k) in shapes:
            lv = _Level()
            lv.H, lv.W, lv.K, lv.Kc = h, w, k, np.array(k, dtype=np.float64)
            maps = []
            for _ in range(6):
                maps.append(flat[off:off + h * w].view(h, w))
                off += h * w
            lv.I, lv.D, lv.gIx, lv.gIy, lv.gDx, lv.gDy = maps
            levels.append(lv)
        self._last_flat = flat
        return levels

    def _build(self, levels: List[_Level], color: torch.Tensor, depth: torch.Tensor, depth_max: float, gradients: bool = True):
        lib, st = L.load_library(), L.stream_ptr()
        l0 = levels[0]
        L.check(lib.bs_odo_prepare(L.p(color), L.p(depth), 1, l0.H, l0.W, float(depth_max), L.p(l0.I), L.p(l0.D), st), "bs_odo_prepare")
        for p, n in zip(levels[:-1], levels[1:]):
            L.check(lib.bs_odo_pyrdown(L.p(p.I), 1, p.H, p.W, L.p(n.I), 0, 0.0, st), "bs_odo_pyrdown")
            L.check(lib.bs_odo_pyrdown(L.p(p.D), 1, p.H, p.W, L.p(n.D), 1, 2.0 * DEPTH_OUTLIER_TRUNC, st), "bs_odo_pyrdown")
        if gradients:
            for lv in levels:
                L.check(lib.bs_odo_sobel(L.p(lv.I), 1, lv.H, lv.W, L.p(lv.gIx), L.p(lv.gIy), st), "bs_odo_sobel")
                L.check(lib.bs_odo_sobel(L.p(lv.D), 1, lv.H, lv.W, L.p(lv.gDx), L.p(lv.gDy), st), "bs_odo_sobel")

    def _pyramid(self, color_u8, depth_m, depth_max: float, gradients: bool):
        color, depth = self._dev(color_u8, torch.uint8), self._dev(depth_m, torch.float32)      # numpy, host or device tensors
        levels = self._alloc_levels(*depth.shape)
        self._build(levels, color, depth, depth_max, gradients)
        return levels

    def _steps(self, ps: List[_Level], pt: List[_Level], T_dev: torch.Tensor, partial: torch.Tensor, out: torch.Tensor):
        """the coarse-to-fine loop on the device: sum(iterations) x (sums kernel, fixed-order reduction + 6x6 solve + pose update)"""
        lib, st = L.load_library(), L.stream_ptr()
        for level, iters in zip(range(len(ps) - 1, -1, -1), self.iterations):
            s, t = ps[level], pt[level]
            L.check(lib.bs_odo_step(L.p(s.I), L.p(s.D), L.p(t.I), L.p(t.D), L.p(t.gIx), L.p(t.gIy), L.p(t.gDx), L.p(t.gDy), 1, s.H, s.W,
                                    s.Kc.ctypes.data_as(C.c_void_p), L.p(T_dev), iters, DEPTH_OUTLIER_TRUNC, DEPTH_HUBER, INTENSITY_HUBER,
                                    L.p(partial), L.p(out), self.flags, st), "bs_odo_step")

    # ---- the estimate ----------------------------------------------------------------------------------
    def estimate(self, src_color, src_depth, tgt_color, tgt_depth, depth_max: float, init: Optional[np.ndarray] = None, trace: bool = False):
        """T (4x4 float64): source points -> target frame"""
        lib, st = L.load_library(), L.stream_ptr()
        ps = self._pyramid(src_color, src_depth, depth_max, gradients=False)
        pt = self._pyramid(tgt_color, tgt_depth, depth_max, gradients=True)
        T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
        out = torch.zeros(29, dtype=torch.float64, device=self.dev)
        partial = torch.empty((ps[0].H * ps[0].W + 255) // 256, 29, dtype=torch.float64, device=self.dev)
        if not trace:
            # no host round trip until the pose is read back; trace=True below walks the same steps from the host (tests, diagnostics)
            T_dev = torch.from_numpy(np.ascontiguousarray(T[:3].reshape(12))).to(self.dev)
            self._steps(ps, pt, T_dev, partial, out)
            T[:3] = T_dev.cpu().numpy().reshape(3, 4)
            self.last_trace = None
            return T
        iu = np.triu_indices(6)
        log = []
        for level, iters in zip(range(len(ps) - 1, -1, -1), self.iterations):
            s, t = ps[level], pt[level]
            for _ in range(iters):
                T12 = np.ascontiguousarray(T[:3].reshape(12))
                L.check(lib.bs_odo_accumulate(L.p(s.I), L.p(s.D), L.p(t.I), L.p(t.D), L.p(t.gIx), L.p(t.gIy), L.p(t.gDx), L.p(t.gDy), s.H, s.W,
                                              s.Kc.ctypes.data_as(C.c_void_p), T12.ctypes.data_as(C.c_void_p), DEPTH_OUTLIER_TRUNC, DEPTH_HUBER,
                                              INTENSITY_HUBER, L.p(partial), L.p(out), self.flags, st), "bs_odo_accumulate")
                r = out.cpu().numpy()                    # (synchronises: T12 / Kc outlive the launch)
                A = np.zeros((6, 6))
                A[iu] = r[:21]
                A = A + np.triu(A, 1).T
                b, res, n = r[21:27], float(r[27]), int(round(r[28]))
                log.append((level, A.copy(), b.copy(), res, n))
                if n < 6:
                    break
                T = se3_exp(np.linalg.solve(A + 1e-12 * np.eye(6), -b)) @ T
        self.last_trace = log
        return T

    def __call__(self, curr_rgbd, prev_rgbd) -> np.ndarray:
        """what VO._compute_vo_o3d returns (visual_odometry.py:97-120): source = the current frame, target = the previous one,
        depth_max = the larger of the two frames' maxima, the estimated transform inverted"""
        dmax = max(self._depth_max(curr_rgbd), self._depth_max(prev_rgbd))
        T = self.estimate(curr_rgbd.color, curr_rgbd.depth, prev_rgbd.color, prev_rgbd.depth, dmax)
        return np.linalg.inv(T)

    # ---- a stream of consecutive frames --------------------------------------------------------------
    def reset(self):
        """forget the previous frame of track()"""
        for k in (self._trk, self._blk):
            if k is not None:
                k["have_prev"] = False

    def track(self, color_u8: torch.Tensor, depth_m: torch.Tensor, graph: bool = True) -> Optional[torch.Tensor]:
        """The next frame of a sequence (device tensors: color uint8 [H, W, 3], depth fp32 metres [H, W]).  Returns None for the first
        frame, then a fresh device tensor: the 12 doubles (rows 0..2) of T(current -> previous), as ``estimate(current, previous)``
        gives them -- nothing is read back to the host here.  depth_max of the pair is max(depth of either frame)
        (visual_odometry.py:99), which invalidates nothing, so the frames are prepared once and the previous frame's pyramid is
        reused as the target."""
        assert color_u8.is_cuda and depth_m.is_cuda and color_u8.dtype == torch.uint8 and depth_m.dtype == torch.float32
        H, W = depth_m.shape
        k = self._trk
        if k is None or k["hw"] != (H, W):
            prev = self._alloc_levels(H, W)
            prev_flat = self._last_flat
            cur = self._alloc_levels(H, W)
            cur_flat = self._last_flat
            k = self._trk = dict(hw=(H, W), prev=prev, cur=cur, prev_flat=prev_flat, cur_flat=cur_flat, have_prev=False, graph=None, warm=False,
                                 color=torch.empty(H, W, 3, dtype=torch.uint8, device=self.dev), depth=torch.empty(H, W, device=self.dev),
                                 T=torch.zeros(12, dtype=torch.float64, device=self.dev),
                                 T0=torch.tensor([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=torch.float64, device=self.dev),
                                 partial=torch.empty((H * W + 255) // 256, 29, dtype=torch.float64, device=self.dev),
                                 out=torch.zeros(29, dtype=torch.float64, device=self.dev))
        k["color"].copy_(color_u8)
        k["depth"].copy_(depth_m)
        if not k["have_prev"]:
            self._build(k["prev"], k["color"], k["depth"], 3.0e38)
            k["have_prev"] = True
            return None

        def pair():
            k["T"].copy_(k["T0"])
            self._build(k["cur"], k["color"], k["depth"], 3.0e38)
            self._steps(k["cur"], k["prev"], k["T"], k["partial"], k["out"])
            k["prev_flat"].copy_(k["cur_flat"])                 # the current frame becomes the next pair's target

        if not graph:
            pair()
        elif k["graph"] is None and not k["warm"]:
            pair()                                              # first pair eagerly (lazy one-off initialisations), the second is captured
            k["warm"] = True
        elif k["graph"] is None:
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device=self.dev)
            cap.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    pair()
            torch.cuda.current_stream(self.dev).wait_stream(cap)
            k["graph"] = g
            g.replay()
        else:
            k["graph"].replay()
        return k["T"].clone()

    # ---- a block of consecutive frames at once ---------------------------------------------------------
    def track_block(self, colors_u8: torch.Tensor, depths_m: torch.Tensor, max_block: int = 64) -> torch.Tensor:
        """The next n frames of a sequence (device tensors: colors uint8 [n, H, W, 3], depths fp32 metres [n, H, W]).  Returns a device
        tensor [pairs, 12] (float64): rows 0..2 of T(frame -> the frame before it) for every frame that has a predecessor -- n - 1 pairs
        for the first block after ``reset()`` / construction, n afterwards (the last frame of a block is kept as the next block's
        first target).  Nothing is read back to the host.  Pair for pair bit-equal to ``estimate(current, previous)``."""
        assert colors_u8.is_cuda and depths_m.is_cuda and colors_u8.dtype == torch.uint8 and depths_m.dtype == torch.float32
        n, H, W = depths_m.shape
        if n > max_block:                        # bounded scratch: long inputs go through in pieces
            return torch.cat([self.track_block(colors_u8[a:a + max_block], depths_m[a:a + max_block], max_block) for a in range(0, n, max_block)])
        k = self._blk
        if k is None or k["hw"] != (H, W) or k["slots"] < n + 1:
            old = k if (k is not None and k["hw"] == (H, W) and k["have_prev"]) else None      # a larger block than before: keep the previous frame
            slots = max(n, max_block if n > 1 else 1) + 1
            levels, (h, w, kk) = [], (H, W, self.K)
            for _ in range(len(self.iterations)):
                lv = _Level()
                lv.H, lv.W, lv.K, lv.Kc = h, w, kk, np.array(kk, dtype=np.float64)
                lv.I, lv.D, lv.gIx, lv.gIy, lv.gDx, lv.gDy = (torch.empty(slots, h, w, device=self.dev, dtype=torch.float32) for _ in range(6))
                levels.append(lv)
                h, w, kk = (h + 1) // 2, (w + 1) // 2, tuple(v / 2.0 for v in kk)
            nblk = min((H * W + 255) // 256, 256)
            k = self._blk = dict(hw=(H, W), slots=slots, levels=levels, have_prev=False,
                                 T=torch.empty(slots - 1, 12, dtype=torch.float64, device=self.dev),
                                 T0=torch.tensor([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=torch.float64, device=self.dev),
                                 partial=torch.empty(slots - 1, nblk, 29, dtype=torch.float64, device=self.dev),
                                 out=torch.zeros(slots - 1, 29, dtype=torch.float64, device=self.dev))
            if old is not None:      # (a first call with one frame allocates two slots; the next, larger block must still pair with that frame)
                for l_new, l_old in zip(levels, old["levels"]):
                    for name in ("I", "D", "gIx", "gIy", "gDx", "gDy"):
                        getattr(l_new, name)[0].copy_(getattr(l_old, name)[0])
                k["have_prev"] = True
        if n == 0:
            return torch.empty(0, 12, dtype=torch.float64, device=self.dev)
        lib, st = L.load_library(), L.stream_ptr()
        colors_u8, depths_m = colors_u8.contiguous(), depths_m.contiguous()
        lv = k["levels"]
        # slot 0 holds the previous block's last frame; the n new frames go to slots 1..n
        L.check(lib.bs_odo_prepare(L.p(colors_u8), L.p(depths_m), n, H, W, 3.0e38, L.p(lv[0].I[1]), L.p(lv[0].D[1]), st), "bs_odo_prepare")
        for p_, n_ in zip(lv[:-1], lv[1:]):
            L.check(lib.bs_odo_pyrdown(L.p(p_.I[1]), n, p_.H, p_.W, L.p(n_.I[1]), 0, 0.0, st), "bs_odo_pyrdown")
            L.check(lib.bs_odo_pyrdown(L.p(p_.D[1]), n, p_.H, p_.W, L.p(n_.D[1]), 1, 2.0 * DEPTH_OUTLIER_TRUNC, st), "bs_odo_pyrdown")
        for l_ in lv:                           # every new frame is a target, in this block or (the last one) in the next
            L.check(lib.bs_odo_sobel(L.p(l_.I[1]), n, l_.H, l_.W, L.p(l_.gIx[1]), L.p(l_.gIy[1]), st), "bs_odo_sobel")
            L.check(lib.bs_odo_sobel(L.p(l_.D[1]), n, l_.H, l_.W, L.p(l_.gDx[1]), L.p(l_.gDy[1]), st), "bs_odo_sobel")
        first = 0 if k["have_prev"] else 1       # the first pair's target slot
        pairs = n - first
        T = k["T"][:max(pairs, 0)]
        if pairs > 0:
            T.copy_(k["T0"].expand(pairs, 12))
            for level, iters in zip(range(len(lv) - 1, -1, -1), self.iterations):
                l_ = lv[level]
                s, t = first + 1, first
                L.check(lib.bs_odo_step(L.p(l_.I[s]), L.p(l_.D[s]), L.p(l_.I[t]), L.p(l_.D[t]), L.p(l_.gIx[t]), L.p(l_.gIy[t]), L.p(l_.gDx[t]),
                                        L.p(l_.gDy[t]), pairs, l_.H, l_.W, l_.Kc.ctypes.data_as(C.c_void_p), L.p(T), iters, DEPTH_OUTLIER_TRUNC,
                                        DEPTH_HUBER, INTENSITY_HUBER, L.p(k["partial"]), L.p(k["out"]), self.flags, st), "bs_odo_step")
        for l_ in lv:                           # the block's last frame is the next block's first target
            for m in (l_.I, l_.D, l_.gIx, l_.gIy, l_.gDx, l_.gDy):
                m[0].copy_(m[n])
        k["have_prev"] = True
        return T.clone()

    def _dev(self, x, dtype) -> torch.Tensor:
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
        return t.to(device=self.dev, dtype=dtype).contiguous()

    @staticmethod
    def _depth_max(rgbd) -> float:
        m = getattr(rgbd, "depth_max", None)
        if m is not None:
            return float(m)
        d = rgbd.depth
        return float(d.max()) if isinstance(d, torch.Tensor) else float(np.nanmax(np.asarray(d)))
