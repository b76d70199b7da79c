"""Analytic RGB-D renderer for the odometry tests: a textured height field Z = g(X, Y) seen by a pinhole camera at a known pose.
Every pixel's ray is intersected with the surface by Newton iterations, so depth and colour of both views are exact (up to u8
colour quantisation) and the relative motion between two renderings is known."""
import numpy as np


def g(X, Y):
    return 0.30 + 0.03 * np.sin(9.0 * X) * np.cos(7.0 * Y) + 0.02 * np.cos(5.0 * X + 3.0 * Y)


def texture(X, Y):
    f = lambda a, b, c: 0.5 + 0.25 * np.sin(a * X + c) * np.cos(b * Y - c) + 0.2 * np.sin((a + b) * (X - Y) + 2 * c)
    return np.stack([f(60, 45, 0.3), f(40, 70, 1.1), f(75, 30, 2.0)], -1)


def render(pose, K, H, W):
    """pose: camera -> world 4x4.  Returns (colour u8 [H, W, 3], depth fp32 [H, W] = camera-frame z of the surface point)"""
    fx, fy, cx, cy = K
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u, dtype=np.float64)], -1)       # ray with unit camera z
    R, o = pose[:3, :3], pose[:3, 3]
    d = d_cam @ R.T
    s = np.full((H, W), 0.3)
    eps = 1e-6
    for _ in range(30):                                                     # Newton on F(s) = (o + s d).z - g((o + s d).x, (o + s d).y)
        P = o + s[..., None] * d
        F = P[..., 2] - g(P[..., 0], P[..., 1])
        gx = (g(P[..., 0] + eps, P[..., 1]) - g(P[..., 0] - eps, P[..., 1])) / (2 * eps)
        gy = (g(P[..., 0], P[..., 1] + eps) - g(P[..., 0], P[..., 1] - eps)) / (2 * eps)
        s = s - F / (d[..., 2] - gx * d[..., 0] - gy * d[..., 1])
    P = o + s[..., None] * d
    col = np.clip(np.rint(texture(P[..., 0], P[..., 1]) * 255.0), 0, 255).astype(np.uint8)
    return col, s.astype(np.float32)                                        # (unit camera z per ray: s IS the camera-frame depth)


def small_pose(rx, ry, rz, tx, ty, tz):
    cxr, sxr, cyr, syr, czr, szr = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cxr, -sxr], [0, sxr, cxr]])
    Ry = np.array([[cyr, 0, syr], [0, 1, 0], [-syr, 0, cyr]])
    Rz = np.array([[czr, -szr, 0], [szr, czr, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = (tx, ty, tz)
    return T
