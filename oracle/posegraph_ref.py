"""CPU oracle for the pose-graph step -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file.  The reference's PoseGraph (BodySLAM_not_refactored/3DM/posegraph.py:5-43) is a thin
wrapper over Open3D: ``o3d.pipelines.registration.global_optimization(pose_graph, GlobalOptimizationLevenbergMarquardt(),
GlobalOptimizationConvergenceCriteria(), GlobalOptimizationOption(max_correspondence_distance=0.005, edge_prune_threshold=0.05,
preference_loop_closure=0.01, reference_node=0))``, called every 500 frames on the chain's nodes and odometry edges
(3DM/slam.py:156-175).  Open3D is a third-party C++ dependency that is not vendored under /root/reference and not installed
here: **parity unpinned**.  This file restates the published algorithm of Open3D's
pipelines/registration/GlobalOptimization.cpp (Choi, Zhou, Koltun, "Robust Reconstruction of Indoor Scenes", CVPR 2015: pose
graph with line processes) in plain, loop-based, dense numpy -- deliberately structured differently from the product
(bodyslam_amd/posegraph.py: vectorised over edges, sparse normal equations) so that the two check each other:

  residual of edge (s, t, X):  r = lin6(X^-1 Tt^-1 Ts),  lin6(M) = [(M21-M12)/2, (M02-M20)/2, (M10-M01)/2, M03, M13, M23]
  Jacobians (left perturbation exp(d) T, generators G_i):  Js[:, i] = lin6(X^-1 Tt^-1 G_i Ts),  Jt[:, i] = -Js[:, i]
  cost  sum_e l_e r_e^T L_e r_e + sum_{uncertain} mu (sqrt(l_e) - 1)^2,  l_e = 1 for certain edges, else (mu / (mu + r^T L r))^2
  mu = preference_loop_closure * max_correspondence_distance^2 * mean_{uncertain} L_e[5, 5]
  Levenberg-Marquardt on the 6N unknowns with the reference node fixed; update  T_i <- exp6(delta_i) T_i,
  exp6(d) = [Rz(d2) Ry(d1) Rx(d0) | d3:6];  stopping rules and defaults of GlobalOptimizationConvergenceCriteria.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Criteria:      # GlobalOptimizationConvergenceCriteria defaults
    max_iteration: int = 100
    min_relative_increment: float = 1e-6
    min_relative_residual_increment: float = 1e-6
    min_right_term: float = 1e-6
    min_residual: float = 1e-6
    max_iteration_lm: int = 20
    upper_scale_factor: float = 2.0 / 3.0
    lower_scale_factor: float = 1.0 / 3.0


def generators():
    G = np.zeros((6, 4, 4))
    G[0, 1, 2], G[0, 2, 1] = -1, 1
    G[1, 2, 0], G[1, 0, 2] = -1, 1
    G[2, 0, 1], G[2, 1, 0] = -1, 1
    G[3, 0, 3] = G[4, 1, 3] = G[5, 2, 3] = 1
    return G


def lin6(M):
    return np.array([(M[2, 1] - M[1, 2]) / 2, (M[0, 2] - M[2, 0]) / 2, (M[1, 0] - M[0, 1]) / 2, M[0, 3], M[1, 3], M[2, 3]])


def exp6(d):
    cx, sx, cy, sy, cz, sz = np.cos(d[0]), np.sin(d[0]), np.cos(d[1]), np.sin(d[1]), np.cos(d[2]), np.sin(d[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = d[3:6]
    return T


def optimize(nodes, edges, transforms, infos, uncertain, max_correspondence_distance=0.005, edge_prune_threshold=0.05,
             preference_loop_closure=0.01, reference_node=0, criteria: Criteria = Criteria()):
    """nodes [N,4,4] f64; edges [(source, target)]; transforms [E,4,4]; infos [E,6,6]; uncertain [E] bool
    -> (optimised nodes [N,4,4], line process weights [E], kept edge mask [E], log dict)."""
    X = [np.array(n, dtype=np.float64) for n in nodes]
    N, E = len(X), len(edges)
    G = generators()
    unc = np.asarray(uncertain, dtype=bool)
    mu = 0.0
    if unc.any():
        mu = preference_loop_closure * max_correspondence_distance ** 2 * float(np.mean([infos[e][5, 5] for e in range(E) if unc[e]]))
    l = np.ones(E)

    def zeta(P):
        out = []
        for e, (s, t) in enumerate(edges):
            out.append(lin6(np.linalg.inv(transforms[e]) @ np.linalg.inv(P[t]) @ P[s]))
        return out

    def total(z, lw):
        r = 0.0
        for e in range(E):
            q = float(z[e] @ infos[e] @ z[e])
            r += lw[e] * q
            if unc[e]:
                r += mu * (np.sqrt(lw[e]) - 1.0) ** 2
        return r

    def line_process(z):
        lw = np.ones(E)
        for e in range(E):
            if unc[e]:
                q = float(z[e] @ infos[e] @ z[e])
                lw[e] = (mu / (mu + q)) ** 2
        return lw

    def system(P, z, lw):
        H = np.zeros((6 * N, 6 * N))
        b = np.zeros(6 * N)
        for e, (s, t) in enumerate(edges):
            A = np.linalg.inv(transforms[e]) @ np.linalg.inv(P[t])
            Js = np.stack([lin6(A @ G[i] @ P[s]) for i in range(6)], axis=1)
            Jt = -Js
            L = lw[e] * infos[e]
            for (a, Ja) in ((s, Js), (t, Jt)):
                b[6 * a:6 * a + 6] -= Ja.T @ L @ z[e]
                for (c, Jc) in ((s, Js), (t, Jt)):
                    H[6 * a:6 * a + 6, 6 * c:6 * c + 6] += Ja.T @ L @ Jc
        rr = slice(6 * reference_node, 6 * reference_node + 6)
        H[rr, :] = 0
        H[:, rr] = 0
        H[rr, rr] = np.eye(6)
        b[rr] = 0
        return H, b

    z = zeta(X)
    l = line_process(z) if unc.any() else l
    cur = total(z, l)
    H, b = system(X, z, l)
    lam = 1e-5 * float(np.max(np.diag(H)))
    ni, rho = 2.0, 0.0
    stop = float(np.max(b)) <= criteria.min_right_term or cur < criteria.min_residual
    log = dict(iterations=0, residual0=cur)
    it = 0
    while it < criteria.max_iteration and not stop:
        it += 1
        lm = 0
        while True:
            delta = np.linalg.solve(H + lam * np.eye(6 * N), b)
            xnorm = np.sqrt(sum(float(lin6(P) @ lin6(P)) for P in X))     # a scale of the unknowns for the relative step test
            if np.linalg.norm(delta) <= criteria.min_relative_increment * (xnorm + criteria.min_relative_increment):
                stop = True
            if not stop:
                Xn = [exp6(delta[6 * i:6 * i + 6]) @ X[i] for i in range(N)]
                zn = zeta(Xn)
                new = total(zn, l)
                rho = (cur - new) / (float(delta @ (lam * delta + b)) + 1e-3)
                if rho > 0:
                    if cur - new < criteria.min_relative_residual_increment * cur:
                        stop = True
                    alpha = 1.0 - (2.0 * rho - 1.0) ** 3
                    alpha = min(alpha, criteria.upper_scale_factor)
                    lam *= max(criteria.lower_scale_factor, alpha)
                    ni = 2.0
                    X, z = Xn, zn
                    if unc.any():
                        l = line_process(z)
                    cur = total(z, l)
                    H, b = system(X, z, l)
                    if float(np.max(b)) <= criteria.min_right_term:
                        stop = True
                else:
                    lam *= ni
                    ni *= 2.0
            lm += 1
            if lm > criteria.max_iteration_lm:
                stop = True
            if rho > 0 or stop:
                break
        if cur < criteria.min_residual:
            stop = True
    log.update(iterations=it, residual=cur)
    keep = np.array([(not unc[e]) or l[e] >= edge_prune_threshold for e in range(E)])
    return np.stack(X), l, keep, log
