"""CPU oracle for the dense RGB-D odometry of the VO step -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file.  The reference (BodySLAM_not_refactored/3DM/visual_odometry.py:97-120) calls Open3D's tensor
``rgbd_odometry_multi_scale(source = current frame, target = previous frame, intrinsic, init, depth_scale = 1000, depth_max,
[20, 10, 5] iterations, Method.Hybrid)`` and inverts the result.  Open3D is a third-party C++ / CUDA dependency that is neither
vendored under /root/reference nor installed here: **parity unpinned**, and more loosely than for the TSDF or the pose graph --
Open3D's preprocessing, pyramid filters and loss parameters are restated here from the published structure of its hybrid odometry
(Park, Zhou, Koltun, "Colored Point Cloud Registration Revisited", ICCV 2017; Steinbruecker et al. 2011), not from its source:

  per frame:   intensity = (0.299 R + 0.587 G + 0.114 B) / 255;  depth in metres, <= 0 or > depth_max -> NaN
  pyramid:     3 levels, level l is level l-1 filtered with the separable [1 4 6 4 1]/16 kernel and subsampled by 2 (depth: the same
               weights over the valid pixels whose depth is within 2 * depth_outlier_trunc of the centre, NaN if the centre is
               invalid); intrinsics halved per level; coarse to fine with 20 / 10 / 5 iterations
  per level:   target Sobel gradients of intensity and depth (3x3, scaled by 1/8; NaN where a depth neighbour is invalid)
  iteration:   for every valid source pixel: p = T v_s;  (u, v) = projection of p.
               association="nearest" (Open3D's, the product's default): the target images and gradients are read at the pixel
               (round(u), round(v)), round half away from zero; skip outside the image / invalid depth or depth gradient there /
               |D_t - p.z| > depth_outlier_trunc (0.07);  r_I = I_t - I_s,  r_D = D_t - p.z;
               loss="o3d": (sum J^T J) delta = - sum J^T huber'(r) with huber'(r) = r clipped to +-delta (0.1 intensity, 0.05 depth),
               J^T J unweighted; cost = sum huber(r).
               association="bilinear" (round 2's variant, an option): the four neighbours are interpolated (any invalid one skips the
               pixel) and loss="irls" weights both sides with w = min(1, delta / |r|): a smooth cost -- the nearest-pixel cost is
               piecewise constant, and its fixed point sits up to half a pixel from the true motion.
               J_I, J_D = the derivatives of the two residuals w.r.t. a left twist (omega, nu) of T (formulas in `_jac`);
               T <- exp(delta) T   (Open3D composes Euler angles instead of the exponential: equal to first order, same fixed point;
               its early stop on relative fitness / rmse changes of 1e-6 is not restated: all 20 / 10 / 5 iterations run)
  result:      T maps source points into the target frame (the reference then inverts it).
The product (bodyslam_amd/rgbd_odometry.py + csrc/odometry.hip) implements exactly this statement; tests compare the two per step
and check both against rendered ground-truth motion."""
from __future__ import annotations

import numpy as np

K5 = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0
DEPTH_OUTLIER_TRUNC, DEPTH_HUBER, INTENSITY_HUBER = 0.07, 0.05, 0.1
ITERATIONS = (20, 10, 5)           # coarse -> fine


def prepare(color_u8, depth_m, depth_max):
    c = np.asarray(color_u8, dtype=np.float64)
    inten = (0.299 * c[..., 0] + 0.587 * c[..., 1] + 0.114 * c[..., 2]) / 255.0
    d = np.asarray(depth_m, dtype=np.float64).copy()
    d[~((d > 0) & (d <= depth_max))] = np.nan
    return inten, d


def _taps(img, pad_value):
    """the 5x5 neighbourhood of every even pixel: [5, 5, h2, w2]"""
    H, W = img.shape
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    p = np.full((H + 4, W + 4), pad_value, dtype=np.float64)
    p[2:2 + H, 2:2 + W] = img
    p[:2, :], p[2 + H:, :] = p[2:3, :], p[1 + H:2 + H, :]                  # replicate borders
    p[:, :2], p[:, 2 + W:] = p[:, 2:3], p[:, 1 + W:2 + W]
    return np.stack([np.stack([p[dy:dy + 2 * h2:2, dx:dx + 2 * w2:2] for dx in range(5)]) for dy in range(5)])


def pyr_down(img):
    t = _taps(img, 0.0)
    return np.einsum("i,j,ijhw->hw", K5, K5, t)


def pyr_down_depth(d, thr=2 * DEPTH_OUTLIER_TRUNC):
    t = _taps(d, np.nan)
    centre = t[2, 2]
    w = K5[:, None, None, None] * K5[None, :, None, None] * (np.abs(t - centre[None, None]) <= thr)     # NaN comparisons are False
    s = np.where(w > 0, t, 0.0)
    out = (w * s).sum((0, 1)) / np.maximum(w.sum((0, 1)), 1e-30)
    out[np.isnan(centre)] = np.nan
    return out


def sobel(img):
    """(d/dx, d/dy) with the 3x3 Sobel kernels scaled by 1/8, replicate borders; NaN propagates"""
    p = np.pad(img, 1, mode="edge")
    gx = ((p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])) / 8.0
    gy = ((p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])) / 8.0
    return gx, gy


def se3_exp(delta):
    w, v = np.asarray(delta[:3], dtype=np.float64), np.asarray(delta[3:], dtype=np.float64)
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        R, V = np.eye(3) + Wx, np.eye(3) + 0.5 * Wx
    else:
        a, b, c = np.sin(th) / th, (1 - np.cos(th)) / th ** 2, (th - np.sin(th)) / th ** 3
        R = np.eye(3) + a * Wx + b * Wx @ Wx
        V = np.eye(3) + b * Wx + c * Wx @ Wx
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, V @ v
    return T


def accumulate(Is, Ds, It, Dt, grads, K, T, association="bilinear", loss=None):
    """(A [6, 6], b [6], residual, inliers) of one Gauss-Newton step at pose T (source -> target)"""
    loss = loss or ("o3d" if association == "nearest" else "irls")
    fx, fy, cx, cy = K
    H, W = Ds.shape
    dIx, dIy, dDx, dDy = grads
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    ok = ~np.isnan(Ds)
    z = np.where(ok, Ds, 1.0)
    X, Y, Z = (u - cx) * z / fx, (v - cy) * z / fy, z
    px = T[0, 0] * X + T[0, 1] * Y + T[0, 2] * Z + T[0, 3]
    py = T[1, 0] * X + T[1, 1] * Y + T[1, 2] * Z + T[1, 3]
    pz = T[2, 0] * X + T[2, 1] * Y + T[2, 2] * Z + T[2, 3]
    ok &= pz > 0
    pzs = np.where(ok, pz, 1.0)
    uf, vf = fx * px / pzs + cx, fy * py / pzs + cy
    if association == "nearest":
        rnd = lambda a: np.sign(a) * np.floor(np.abs(a) + 0.5)          # round half away from zero (roundf)
        ur, vr = rnd(np.where(ok, uf, 0.0)), rnd(np.where(ok, vf, 0.0))
        ok &= (ur >= 0) & (ur <= W - 1) & (vr >= 0) & (vr <= H - 1)
        ui, vi = np.where(ok, ur, 0).astype(int), np.where(ok, vr, 0).astype(int)

        def bil(img):    # the nearest pixel
            return img[vi, ui]
    else:
        ok &= (uf >= 0) & (uf <= W - 1) & (vf >= 0) & (vf <= H - 1)
        uf, vf = np.where(ok, uf, 0.0), np.where(ok, vf, 0.0)
        u0, v0 = np.minimum(np.floor(uf).astype(int), W - 2), np.minimum(np.floor(vf).astype(int), H - 2)
        au, av = uf - u0, vf - v0

        def bil(img):    # bilinear sample at (vf, uf); NaN if any of the four neighbours is NaN
            return ((1 - av) * ((1 - au) * img[v0, u0] + au * img[v0, u0 + 1]) + av * ((1 - au) * img[v0 + 1, u0] + au * img[v0 + 1, u0 + 1]))

    dt = bil(Dt)
    rD = dt - pz
    gx, gy, hx, hy = bil(dIx), bil(dIy), bil(dDx), bil(dDy)
    ok &= ~np.isnan(dt) & ~np.isnan(hx) & ~np.isnan(hy) & (np.abs(np.where(np.isnan(rD), 1e9, rD)) <= DEPTH_OUTLIER_TRUNC)
    rI = bil(It) - Is
    JI, JD = _jac(px, py, pzs, gx, gy, hx, hy, fx, fy)
    wI = np.where(np.abs(rI) <= INTENSITY_HUBER, 1.0, INTENSITY_HUBER / np.maximum(np.abs(rI), 1e-30))
    wD = np.where(np.abs(rD) <= DEPTH_HUBER, 1.0, DEPTH_HUBER / np.maximum(np.abs(rD), 1e-30))
    m = ok.ravel()
    JI, JD = JI.reshape(6, -1)[:, m], JD.reshape(6, -1)[:, m]
    rI, rD, wI, wD = rI.ravel()[m], rD.ravel()[m], wI.ravel()[m], wD.ravel()[m]
    if loss == "o3d":
        qI = np.where(np.abs(rI) < INTENSITY_HUBER, rI, np.sign(rI) * INTENSITY_HUBER)
        qD = np.where(np.abs(rD) < DEPTH_HUBER, rD, np.sign(rD) * DEPTH_HUBER)
        hub = lambda r, d: np.where(np.abs(r) < d, 0.5 * r * r, d * (np.abs(r) - 0.5 * d))
        return JI @ JI.T + JD @ JD.T, JI @ qI + JD @ qD, float((hub(rI, INTENSITY_HUBER) + hub(rD, DEPTH_HUBER)).sum()), int(m.sum())
    A = (JI * wI) @ JI.T + (JD * wD) @ JD.T
    b = (JI * wI) @ rI + (JD * wD) @ rD
    return A, b, float((wI * rI * rI + wD * rD * rD).sum()), int(m.sum())


def _jac(X, Y, Z, gx, gy, hx, hy, fx, fy):
    """rows of d r_I / d(omega, nu) and d r_D / d(omega, nu) for the transformed point (X, Y, Z): a left twist moves it by
    omega x p + nu; the pixel moves by the projection's derivative, the residuals by the target gradients at the pixel"""
    iz = 1.0 / Z
    c0, c1 = gx * fx * iz, gy * fy * iz
    c2 = -(c0 * X + c1 * Y) * iz
    d0, d1 = hx * fx * iz, hy * fy * iz
    d2 = -(d0 * X + d1 * Y) * iz
    JI = np.stack([-Z * c1 + Y * c2, Z * c0 - X * c2, -Y * c0 + X * c1, c0, c1, c2])
    JD = np.stack([(-Z * d1 + Y * d2) - Y, (Z * d0 - X * d2) + X, -Y * d0 + X * d1, d0, d1, d2 - 1.0])
    return JI, JD


def build_pyramid(inten, depth, K, levels=3):
    out = [(inten, depth, tuple(K))]
    for _ in range(levels - 1):
        i, d, k = out[-1]
        out.append((pyr_down(i), pyr_down_depth(d), (k[0] / 2, k[1] / 2, k[2] / 2, k[3] / 2)))
    return out


def rgbd_odometry(src_color, src_depth, tgt_color, tgt_depth, K, depth_max, init=None, iterations=ITERATIONS, trace=None,
                  association="bilinear", loss=None):
    """T (4x4): source points -> target frame"""
    ps = build_pyramid(*prepare(src_color, src_depth, depth_max), K)
    pt = build_pyramid(*prepare(tgt_color, tgt_depth, depth_max), K)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
    for level, iters in zip(range(len(ps) - 1, -1, -1), iterations):
        Is, Ds, k = ps[level]
        It, Dt, _ = pt[level]
        grads = (*sobel(It), *sobel(Dt))
        for _ in range(iters):
            A, b, res, n = accumulate(Is, Ds, It, Dt, grads, k, T, association, loss)
            if trace is not None:
                trace.append((level, A.copy(), b.copy(), res, n))
            if n < 6:
                break
            delta = np.linalg.solve(A + 1e-12 * np.eye(6), -b)
            T = se3_exp(delta) @ T
    return T
