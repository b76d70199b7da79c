"""Pose graph of the SLAM loop: drop-in for BodySLAM_not_refactored/3DM/posegraph.py:5-43.

The reference keeps an Open3D ``PoseGraph`` of the chain's absolute poses with one odometry edge per frame
(3DM/slam.py:156-157: ``add_node(curr_absolute_pose)``, ``add_edge(transformation, i, i - 1, False)``), runs Open3D's
Levenberg-Marquardt global optimisation every 500 frames (:159-165) and reads the node poses back
(``update_global_extrinsic``, 3DM/slam_utils.py:88-90).  Same class, method names, argument meaning and defaults here; the
optimiser is a restatement of Open3D's GlobalOptimization (Choi, Zhou, Koltun, CVPR 2015: pose graph with line processes,
see oracle/posegraph_ref.py for the formulas) on the host, as in the reference: edges are linearised in one vectorised pass
(fp64), the 6N x 6N normal equations are assembled block-sparse (block tridiagonal for a chain, plus one off-diagonal block
pair per loop closure) and solved by a sparse LU per LM step -- 4 000 nodes (BASELINE config 5) take milliseconds per step
where a dense solve would need 4.6 GB.  With odometry edges only, the chain itself is the optimum: the residual is below
``min_residual`` at entry and the poses are returned untouched (what the reference's log line "posegraph non fa nulla" reports).
Open3D is not vendored with the reference and not installable offline: parity against it is unpinned; tests/ check this
module against the independent dense oracle and against the chain's golden vectors.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class PoseGraphNode:
    pose: np.ndarray


@dataclass
class PoseGraphEdge:
    source_node_id: int
    target_node_id: int
    transformation: np.ndarray
    information: np.ndarray
    uncertain: bool
    weight: float = 1.0          # line process value after optimize() (Open3D keeps it in confidence_)


@dataclass
class _Graph:
    nodes: List[PoseGraphNode] = field(default_factory=list)
    edges: List[PoseGraphEdge] = field(default_factory=list)


@dataclass
class ConvergenceCriteria:       # o3d GlobalOptimizationConvergenceCriteria() defaults
    max_iteration: int = 100
    min_relative_increment: float = 1e-6
    min_relative_residual_increment: float = 1e-6
    min_right_term: float = 1e-6
    min_residual: float = 1e-6
    max_iteration_lm: int = 20
    upper_scale_factor: float = 2.0 / 3.0
    lower_scale_factor: float = 1.0 / 3.0


_GEN = np.zeros((6, 4, 4))
_GEN[0, 1, 2], _GEN[0, 2, 1] = -1, 1
_GEN[1, 2, 0], _GEN[1, 0, 2] = -1, 1
_GEN[2, 0, 1], _GEN[2, 1, 0] = -1, 1
_GEN[3, 0, 3] = _GEN[4, 1, 3] = _GEN[5, 2, 3] = 1


def _lin6(M: np.ndarray) -> np.ndarray:
    """[..., 4, 4] -> [..., 6]: the linearised 6-vector of a near-identity transform"""
    return np.stack([(M[..., 2, 1] - M[..., 1, 2]) / 2, (M[..., 0, 2] - M[..., 2, 0]) / 2, (M[..., 1, 0] - M[..., 0, 1]) / 2,
                     M[..., 0, 3], M[..., 1, 3], M[..., 2, 3]], axis=-1)


def _exp6(d: np.ndarray) -> np.ndarray:
    """[N, 6] -> [N, 4, 4]: Rz(d2) Ry(d1) Rx(d0) and the translation d3:6"""
    cx, sx, cy, sy, cz, sz = np.cos(d[:, 0]), np.sin(d[:, 0]), np.cos(d[:, 1]), np.sin(d[:, 1]), np.cos(d[:, 2]), np.sin(d[:, 2])
    T = np.zeros((d.shape[0], 4, 4))
    T[:, 0, 0], T[:, 0, 1], T[:, 0, 2] = cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx
    T[:, 1, 0], T[:, 1, 1], T[:, 1, 2] = sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx
    T[:, 2, 0], T[:, 2, 1], T[:, 2, 2] = -sy, cy * sx, cy * cx
    T[:, :3, 3] = d[:, 3:6]
    T[:, 3, 3] = 1.0
    return T


def _inv_rigid(T: np.ndarray) -> np.ndarray:
    return np.linalg.inv(T)


class PoseGraph:
    def __init__(self, max_correspondence_distance=0.005, edge_prune_threshold=0.05, preference_loop_closure=0.01,
                 reference_node=0):
        self.pose_graph = _Graph()
        self.max_correspondence_distance = max_correspondence_distance
        self.edge_prune_threshold = edge_prune_threshold
        self.preference_loop_closure = preference_loop_closure
        self.reference_node = reference_node
        self.convergence_criteria = ConvergenceCriteria()
        self.last_log = None

    # ---- the reference's surface -----------------------------------------------------------------
    def add_node(self, extrinsic_matrix):
        self.pose_graph.nodes.append(PoseGraphNode(np.array(self._check_type(extrinsic_matrix), dtype=np.float64)))

    def add_edge(self, motion_matrix, source_id, target_id, uncertain, info=np.eye(6)):
        self.pose_graph.edges.append(PoseGraphEdge(int(source_id), int(target_id), np.array(self._check_type(motion_matrix), dtype=np.float64),
                                                   np.array(self._check_type(info), dtype=np.float64), bool(uncertain)))

    def _check_type(self, matrix):
        if not isinstance(matrix, np.ndarray):
            return matrix.cpu().numpy()
        return matrix

    def optimize(self):
        g = self.pose_graph
        N, E = len(g.nodes), len(g.edges)
        if N == 0 or E == 0:
            return
        for e in g.edges:
            if not (0 <= e.source_node_id < N and 0 <= e.target_node_id < N):
                raise ValueError(f"edge ({e.source_node_id}, {e.target_node_id}) refers to a node outside 0..{N - 1}")
        X = np.stack([n.pose for n in g.nodes])
        src = np.array([e.source_node_id for e in g.edges])
        tgt = np.array([e.target_node_id for e in g.edges])
        Tinv = _inv_rigid(np.stack([e.transformation for e in g.edges]))
        L = np.stack([e.information for e in g.edges])
        unc = np.array([e.uncertain for e in g.edges])
        X, lw, log = self._levenberg_marquardt(X, src, tgt, Tinv, L, unc)
        # edges the line process switched off are dropped (uncertain ones only) and the pruned graph is optimised again: Open3D's
        # global_optimization runs optimise -> prune -> optimise
        keep = (~unc) | (lw >= self.edge_prune_threshold)
        if not keep.all():
            for e, w_ in zip(g.edges, lw):
                e.weight = float(w_)
            g.edges = [e for e, k in zip(g.edges, keep) if k]
            X, lw2, log2 = self._levenberg_marquardt(X, src[keep], tgt[keep], Tinv[keep], L[keep], unc[keep])
            log = log + log2 if isinstance(log, list) and isinstance(log2, list) else log2
            lw = lw2
        self.last_log = log
        for n, P in zip(g.nodes, X):
            n.pose = P
        for e, w_ in zip(g.edges, lw):
            e.weight = float(w_)

    # ---- Levenberg-Marquardt with line processes ------------------------------------------------
    def _levenberg_marquardt(self, X, src, tgt, Tinv, L, unc):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        c = self.convergence_criteria
        N, E = X.shape[0], src.shape[0]
        ref = self.reference_node
        mu = 0.0
        if unc.any():
            mu = self.preference_loop_closure * self.max_correspondence_distance ** 2 * float(L[unc, 5, 5].mean())

        def zeta(P):
            return _lin6(Tinv @ _inv_rigid(P[tgt]) @ P[src])                        # [E, 6]

        def quad(z):
            return np.einsum("ei,eij,ej->e", z, L, z)

        def line_process(z):
            lw = np.ones(E)
            if unc.any():
                q = quad(z)
                lw[unc] = (mu / (mu + q[unc])) ** 2
            return lw

        def total(z, lw):
            return float((lw * quad(z)).sum() + (mu * (np.sqrt(lw[unc]) - 1.0) ** 2).sum())

        rows_s, rows_t = 6 * src[:, None] + np.arange(6), 6 * tgt[:, None] + np.arange(6)           # [E, 6] unknown indices
        keep = np.ones(6 * N)
        keep[6 * ref:6 * ref + 6] = 0.0

        def system(P, z, lw):
            A = Tinv @ _inv_rigid(P[tgt])                                           # [E, 4, 4]
            Js = np.stack([_lin6(A @ _GEN[i] @ P[src]) for i in range(6)], axis=2)  # [E, 6, 6], column i = generator i
            W = lw[:, None, None] * L
            JtW = np.swapaxes(Js, 1, 2) @ W                                         # Js^T (l L)
            Hss = JtW @ Js                                                          # = Htt; Hst = Hts = -Hss  (Jt = -Js)
            g_ = np.einsum("eij,ej->ei", JtW, z)                                    # Js^T (l L) r
            b = np.zeros(6 * N)
            np.add.at(b, rows_s, -g_)
            np.add.at(b, rows_t, g_)
            ri = np.concatenate([np.repeat(rows_s, 6, 1), np.repeat(rows_s, 6, 1), np.repeat(rows_t, 6, 1), np.repeat(rows_t, 6, 1)]).ravel()
            ci = np.concatenate([np.tile(rows_s, (1, 6)), np.tile(rows_t, (1, 6)), np.tile(rows_s, (1, 6)), np.tile(rows_t, (1, 6))]).ravel()
            vals = np.concatenate([Hss.reshape(E, 36), -Hss.reshape(E, 36), -Hss.reshape(E, 36), Hss.reshape(E, 36)]).ravel()
            vals = vals * keep[ri] * keep[ci]                                       # the reference node stays where it is
            H = sp.coo_matrix((vals, (ri, ci)), shape=(6 * N, 6 * N)).tocsc()
            H = H + sp.diags(1.0 - keep)
            return H, b * keep

        z = zeta(X)
        lw = line_process(z)
        cur = total(z, lw)
        H, b = system(X, z, lw)
        lam = 1e-5 * float(H.diagonal().max())
        ni, rho = 2.0, 0.0
        stop = float(b.max()) <= c.min_right_term or cur < c.min_residual
        log = dict(iterations=0, residual0=cur)
        it = 0
        eye = sp.identity(6 * N, format="csc")
        while it < c.max_iteration and not stop:
            it += 1
            lm = 0
            while True:
                delta = spla.splu((H + lam * eye).tocsc()).solve(b)
                xnorm = float(np.sqrt((_lin6(X) ** 2).sum()))
                if np.linalg.norm(delta) <= c.min_relative_increment * (xnorm + c.min_relative_increment):
                    stop = True
                if not stop:
                    Xn = _exp6(delta.reshape(N, 6)) @ X
                    zn = zeta(Xn)
                    new = total(zn, lw)
                    rho = (cur - new) / (float(delta @ (lam * delta + b)) + 1e-3)
                    if rho > 0:
                        if cur - new < c.min_relative_residual_increment * cur:
                            stop = True
                        alpha = min(1.0 - (2.0 * rho - 1.0) ** 3, c.upper_scale_factor)
                        lam *= max(c.lower_scale_factor, alpha)
                        ni = 2.0
                        X, z = Xn, zn
                        lw = line_process(z)
                        cur = total(z, lw)
                        H, b = system(X, z, lw)
                        if float(b.max()) <= c.min_right_term:
                            stop = True
                    else:
                        lam *= ni
                        ni *= 2.0
                lm += 1
                if lm > c.max_iteration_lm:
                    stop = True
                if rho > 0 or stop:
                    break
            if cur < c.min_residual:
                stop = True
        log.update(iterations=it, residual=cur)
        return X, lw, log


def update_global_extrinsic(global_pose_graph) -> list:
    """3DM/slam_utils.py:88-90"""
    return [node.pose for node in global_pose_graph.nodes]
