"""Marching-cubes case table for the TSDF mesh extraction (``TSDF.extract_mesh``; reference: BodySLAM_not_refactored/3DM/tsdf.py:42-52 ->
Open3D ``ScalableTSDFVolume.extract_triangle_mesh``, called on the last frame at 3DM/slam.py:189-193).

The classic 256-entry triangle table is not available offline, so the table is GENERATED here, once at import, by the construction
the classic one comes from: for every sign configuration of the 8 cube corners (bit c of the case = corner c is inside, tsdf < 0) the
iso-surface crosses the cube faces in line segments (marching squares per face), the segments of the six faces close into loops on
the cube's surface, and every loop is triangulated as a fan (from a vertex whose diagonals all run through the cube's interior).  A face whose two diagonal corners are inside is ambiguous; it is always
cut so that the two INSIDE corners are separated.  The rule looks at the face's own four corners only, so the two cubes that share a
face cut it identically: the mesh is watertight by construction (the classic table is not, at some ambiguous faces).  Triangles are
wound so that their normal points from the inside (tsdf < 0) to the outside.

Conventions shared with csrc/tsdf.hip: corner c sits at offset (c & 1, (c >> 1) & 1, (c >> 2) & 1) from the cube's voxel;
edge e joins EDGES[e] = (a, b), a < b, along axis log2(a ^ b); ``TRI_TABLE[case]`` lists edge ids, three per triangle, -1 terminated."""
from __future__ import annotations

import numpy as np

CORNER = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)
EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]          # 12 edges, a < b
EDGE_ID = {e: i for i, e in enumerate(EDGES)}
EDGE_AXIS = [int(np.log2(a ^ b)) for a, b in EDGES]


def _faces():
    out = []
    for d in range(3):
        p, q = [a for a in range(3) if a != d]
        for s in (0, 1):
            cyc = []
            for (u, v) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                off = [0, 0, 0]
                off[d], off[p], off[q] = s, u, v
                cyc.append(off[0] + 2 * off[1] + 4 * off[2])
            out.append(cyc)
    return out


FACES = _faces()


def _edge(a: int, b: int) -> int:
    return EDGE_ID[(min(a, b), max(a, b))]


def case_loops(case: int):
    """the closed loops of cube edges the iso-surface crosses for this sign configuration, each oriented inside -> outside"""
    inside = [(case >> c) & 1 for c in range(8)]
    nbr = {}
    for cyc in FACES:
        cross = [k for k in range(4) if inside[cyc[k]] != inside[cyc[(k + 1) % 4]]]
        segs = []
        if len(cross) == 2:
            segs.append((_edge(cyc[cross[0]], cyc[(cross[0] + 1) % 4]), _edge(cyc[cross[1]], cyc[(cross[1] + 1) % 4])))
        elif len(cross) == 4:             # ambiguous face: every inside corner is cut off on its own
            for j in range(4):
                if inside[cyc[j]]:
                    segs.append((_edge(cyc[(j - 1) % 4], cyc[j]), _edge(cyc[j], cyc[(j + 1) % 4])))
        for e0, e1 in segs:
            nbr.setdefault(e0, []).append(e1)
            nbr.setdefault(e1, []).append(e0)
    assert all(len(v) == 2 for v in nbr.values()), (case, nbr)
    loops, seen = [], set()
    for start in sorted(nbr):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            n0, n1 = nbr[cur]
            nxt = n0 if n0 != prev else n1
            if nxt == start:
                break
            loop.append(nxt)
            seen.add(nxt)
            prev, cur = cur, nxt
        # orientation: the polygon's normal (Newell, nominal mid-edge positions) points away from the inside corners it cuts off
        pts = np.array([(CORNER[EDGES[e][0]] + CORNER[EDGES[e][1]]) * 0.5 for e in loop])
        n = np.zeros(3)
        for i in range(len(pts)):
            p, q = pts[i], pts[(i + 1) % len(pts)]
            n += np.cross(p, q)
        ins = np.array([CORNER[a] if inside[a] else CORNER[b] for a, b in (EDGES[e] for e in loop)], dtype=np.float64)
        if np.dot(n, pts.mean(0) - ins.mean(0)) < 0:
            loop = [loop[0]] + loop[:0:-1]
        loops.append(_fan_origin(loop))
    return loops


def _same_face(e0: int, e1: int) -> bool:
    return any(set(EDGES[e0]) <= set(cyc) and set(EDGES[e1]) <= set(cyc) for cyc in FACES)


def _fan_origin(loop):
    """rotate the loop so that no diagonal of its fan (loop[0], loop[i]) joins two vertices of ONE cube face: such a diagonal would lie
    in the face, where the neighbouring cube may put an edge of its own -- four triangles on one edge.  (Loop edges proper are face
    segments and are matched by the neighbour; diagonals must stay inside the cube.)"""
    n = len(loop)
    for r in range(n):
        rot = loop[r:] + loop[:r]
        if all(not _same_face(rot[0], rot[i]) for i in range(2, n - 1)):
            return rot
    raise AssertionError(f"no face-free fan for loop {loop}")


def _build():
    tris = []
    for case in range(256):
        t = []
        if 0 < case < 255:
            for loop in case_loops(case):
                for i in range(1, len(loop) - 1):
                    t += [loop[0], loop[i], loop[i + 1]]
        tris.append(t)
    width = max(len(t) for t in tris) + 1
    tab = np.full((256, width), -1, dtype=np.int32)
    for case, t in enumerate(tris):
        tab[case, :len(t)] = t
    return tab


TRI_TABLE = _build()                                  # int32 [256, 3 * max_triangles + 1]
N_TRI = ((TRI_TABLE >= 0).sum(1) // 3).astype(np.int32)
