"""Dense RGB-D odometry of the VO step: what ``VO._compute_vo_o3d`` gets from Open3D's
``rgbd_odometry_multi_scale(curr, prev, intrinsic, init, 1000.0, depth_max, [20, 10, 5], Method.Hybrid)``
(BodySLAM_not_refactored/3DM/visual_odometry.py:97-120): the relative pose between two RGB-D frames.

Open3D is not vendored and not installable offline, so this is not a restatement of its source but an implementation of the same
published scheme -- hybrid photometric + geometric Gauss-Newton on a 3-level pyramid, coarse to fine with 20 / 10 / 5 iterations,
Huber losses, 0.07 m depth outlier gate (the defaults of Open3D's OdometryLossParams) -- stated precisely in
oracle/rgbd_odometry_ref.py; parity with Open3D is unpinned.  One deliberate difference: the target images are sampled
bilinearly, not at the nearest pixel -- with nearest-pixel sampling the cost is piecewise constant and the Gauss-Newton steps are
rounding noise at the sub-pixel motions of consecutive endoscopy frames (measured on rendered scenes).

The images, pyramids, gradients, the 29 sums of every Gauss-Newton step, the 6x6 solve and the pose update all run in the kernels of
csrc/odometry.hip: a pair is 35 x 3 launches with no host round trip until the pose is read back.  ``RGBDOdometry()(curr_rgbd, prev_rgbd)`` returns what
``_compute_vo_o3d`` returns: the inverse of the estimated source -> target transform."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib as L

DEPTH_OUTLIER_TRUNC, DEPTH_HUBER, INTENSITY_HUBER = 0.07, 0.05, 0.1      # o3d.t.pipelines.odometry.OdometryLossParams defaults
ITERATIONS = (20, 10, 5)                                                   # visual_odometry.py:103-107, coarse -> fine


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
    __slots__ = ("H", "W", "K", "I", "D", "gIx", "gIy", "gDx", "gDy")


class RGBDOdometry:
    def __init__(self, K: Sequence[float], device: int = 0, iterations: Sequence[int] = ITERATIONS):
        """K = (fx, fy, cx, cy) of the full-resolution frames"""
        self.K = tuple(float(v) for v in K)
        self.dev = torch.device("cuda", device)
        self.iterations = tuple(int(i) for i in iterations)
        L.init(device)
        self.last_trace = None

    # ---- device-side image preparation ---------------------------------------------------------------
    def _pyramid(self, color_u8, depth_m, depth_max: float, gradients: bool):
        lib, st = L.load_library(), L.stream_ptr()
        color, depth = self._dev(color_u8, torch.uint8), self._dev(depth_m, torch.float32)      # numpy, host or device tensors
        H, W = depth.shape
        levels = []
        lv = _Level()
        lv.H, lv.W, lv.K = H, W, self.K
        lv.I, lv.D = torch.empty(H, W, device=self.dev), torch.empty(H, W, device=self.dev)
        L.check(lib.bs_odo_prepare(L.p(color), L.p(depth), H, W, float(depth_max), L.p(lv.I), L.p(lv.D), st), "bs_odo_prepare")
        levels.append(lv)
        for _ in range(len(self.iterations) - 1):
            p = levels[-1]
            n = _Level()
            n.H, n.W, n.K = (p.H + 1) // 2, (p.W + 1) // 2, tuple(v / 2.0 for v in p.K)
            n.I, n.D = torch.empty(n.H, n.W, device=self.dev), torch.empty(n.H, n.W, device=self.dev)
            L.check(lib.bs_odo_pyrdown(L.p(p.I), p.H, p.W, L.p(n.I), 0, 0.0, st), "bs_odo_pyrdown")
            L.check(lib.bs_odo_pyrdown(L.p(p.D), p.H, p.W, L.p(n.D), 1, 2.0 * DEPTH_OUTLIER_TRUNC, st), "bs_odo_pyrdown")
            levels.append(n)
        if gradients:
            for lv in levels:
                lv.gIx, lv.gIy, lv.gDx, lv.gDy = (torch.empty(lv.H, lv.W, device=self.dev) for _ in range(4))
                L.check(lib.bs_odo_sobel(L.p(lv.I), lv.H, lv.W, L.p(lv.gIx), L.p(lv.gIy), st), "bs_odo_sobel")
                L.check(lib.bs_odo_sobel(L.p(lv.D), lv.H, lv.W, L.p(lv.gDx), L.p(lv.gDy), st), "bs_odo_sobel")
        return levels

    # ---- the estimate ----------------------------------------------------------------------------------
    def estimate(self, src_color, src_depth, tgt_color, tgt_depth, depth_max: float, init: Optional[np.ndarray] = None, trace: bool = False):
        """T (4x4 float64): source points -> target frame"""
        lib, st = L.load_library(), L.stream_ptr()
        ps = self._pyramid(src_color, src_depth, depth_max, gradients=False)
        pt = self._pyramid(tgt_color, tgt_depth, depth_max, gradients=True)
        T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
        out = torch.zeros(29, dtype=torch.float64, device=self.dev)
        if not trace:
            # the whole coarse-to-fine loop on the device: 35 x (sums, fixed-order reduction, 6x6 solve + pose update), no host
            # round trip until the pose is read back.  trace=True below walks the same steps from the host (tests, diagnostics).
            partial = torch.empty((ps[0].H * ps[0].W + 255) // 256, 29, dtype=torch.float64, device=self.dev)
            T_dev = torch.from_numpy(np.ascontiguousarray(T[:3].reshape(12))).to(self.dev)
            keep = []
            for level, iters in zip(range(len(ps) - 1, -1, -1), self.iterations):
                s, t = ps[level], pt[level]
                Kl = np.array(s.K, dtype=np.float64)
                keep.append(Kl)
                L.check(lib.bs_odo_step(L.p(s.I), L.p(s.D), L.p(t.I), L.p(t.D), L.p(t.gIx), L.p(t.gIy), L.p(t.gDx), L.p(t.gDy), s.H, s.W,
                                        Kl.ctypes.data_as(C.c_void_p), L.p(T_dev), iters, DEPTH_OUTLIER_TRUNC, DEPTH_HUBER, INTENSITY_HUBER,
                                        L.p(partial), L.p(out), st), "bs_odo_step")
            T[:3] = T_dev.cpu().numpy().reshape(3, 4)
            self.last_trace = None
            return T
        partial = torch.empty((ps[0].H * ps[0].W + 255) // 256, 29, dtype=torch.float64, device=self.dev)
        iu = np.triu_indices(6)
        log = [] if trace else None
        for level, iters in zip(range(len(ps) - 1, -1, -1), self.iterations):
            s, t = ps[level], pt[level]
            Kl = np.array(s.K, dtype=np.float64)
            for _ in range(iters):
                T12 = np.ascontiguousarray(T[:3].reshape(12))
                L.check(lib.bs_odo_accumulate(L.p(s.I), L.p(s.D), L.p(t.I), L.p(t.D), L.p(t.gIx), L.p(t.gIy), L.p(t.gDx), L.p(t.gDy), s.H, s.W,
                                              Kl.ctypes.data_as(C.c_void_p), T12.ctypes.data_as(C.c_void_p), DEPTH_OUTLIER_TRUNC, DEPTH_HUBER,
                                              INTENSITY_HUBER, L.p(partial), L.p(out), st), "bs_odo_accumulate")
                r = out.cpu().numpy()                    # (synchronises: T12 / Kl outlive the launch)
                A = np.zeros((6, 6))
                A[iu] = r[:21]
                A = A + np.triu(A, 1).T
                b, res, n = r[21:27], float(r[27]), int(round(r[28]))
                if log is not None:
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
