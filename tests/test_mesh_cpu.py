"""Marching-cubes case table of the product (bodyslam_amd/marching_cubes.py) and the oracle's table-free restatement
(oracle/tsdf_ref.py extract_triangle_mesh): structural properties a correct extraction must have.  CPU only."""
from collections import Counter

import numpy as np

from bodyslam_amd import marching_cubes as MC


def test_table_structure():
    assert MC.TRI_TABLE.shape[0] == 256 and len(MC.EDGES) == 12 and MC.N_TRI[0] == MC.N_TRI[255] == 0
    assert MC.N_TRI.max() <= 5
    for case in range(1, 255):
        t = MC.TRI_TABLE[case]
        t = t[t >= 0]
        assert len(t) % 3 == 0 and len(t) > 0
        inside = [(case >> c) & 1 for c in range(8)]
        crossing = {e for e, (a, b) in enumerate(MC.EDGES) if inside[a] != inside[b]}
        assert set(t.tolist()) == crossing                      # every crossed edge carries a vertex, no other edge does
        comp = MC.TRI_TABLE[255 - case]
        assert set(comp[comp >= 0].tolist()) == crossing        # the complement cuts the same edges


def test_table_is_watertight_and_oriented_on_a_random_field():
    rng = np.random.default_rng(0)
    n = 10
    f = rng.standard_normal((n, n, n))
    directed = Counter()
    for x in range(n - 1):
        for y in range(n - 1):
            for z in range(n - 1):
                case = sum(1 << c for c in range(8) if f[x + MC.CORNER[c][0], y + MC.CORNER[c][1], z + MC.CORNER[c][2]] < 0)
                t = MC.TRI_TABLE[case]
                for tri in t[t >= 0].reshape(-1, 3):
                    ks, ps = [], []
                    for e in tri:
                        a, b = MC.EDGES[e]
                        pa, pb = np.array((x, y, z)) + MC.CORNER[a], np.array((x, y, z)) + MC.CORNER[b]
                        fa, fb = f[tuple(pa)], f[tuple(pb)]
                        ks.append((tuple(pa), MC.EDGE_AXIS[e]))
                        ps.append(pa + (pb - pa) * (fa / (fa - fb)))
                    for i in range(3):
                        directed[(ks[i], ks[(i + 1) % 3])] += 1
    assert max(directed.values()) == 1                           # consistent winding: no directed edge twice
    for (a, b) in directed:                                      # every interior edge has its opposite: only the volume's faces are open
        if (b, a) not in directed:
            pa, pb = np.array(a[0]), np.array(b[0])
            # both vertices lie in a cube face on the volume's boundary (the plane i = 0 or i = n - 1, neither edge running along i)
            on_border = any(a[1] != i and b[1] != i and pa[i] == pb[i] and pa[i] in (0, n - 1) for i in range(3))
            assert on_border, (a, b)


def test_table_orients_triangles_from_inside_to_outside():
    """a sphere (f = distance - R, inside negative): every triangle's normal points away from the centre, and the mesh is closed"""
    n, c0, R = 14, np.array([6.3, 6.6, 6.1]), 4.2
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).astype(np.float64)
    f = np.linalg.norm(g - c0, axis=-1) - R
    directed, count = Counter(), 0
    for x in range(n - 1):
        for y in range(n - 1):
            for z in range(n - 1):
                case = sum(1 << c for c in range(8) if f[x + MC.CORNER[c][0], y + MC.CORNER[c][1], z + MC.CORNER[c][2]] < 0)
                t = MC.TRI_TABLE[case]
                for tri in t[t >= 0].reshape(-1, 3):
                    ks, ps = [], []
                    for e in tri:
                        a, b = MC.EDGES[e]
                        pa, pb = np.array((x, y, z)) + MC.CORNER[a], np.array((x, y, z)) + MC.CORNER[b]
                        fa, fb = f[tuple(pa)], f[tuple(pb)]
                        ks.append((tuple(pa), MC.EDGE_AXIS[e]))
                        ps.append(pa + (pb - pa) * (fa / (fa - fb)))
                    nrm = np.cross(ps[1] - ps[0], ps[2] - ps[0])
                    assert np.dot(nrm, np.mean(ps, axis=0) - c0) > 0
                    for i in range(3):
                        directed[(ks[i], ks[(i + 1) % 3])] += 1
                    count += 1
    assert count > 300 and max(directed.values()) == 1 and all((b, a) in directed for (a, b) in directed)     # closed, oriented


def test_oracle_mesh_of_a_smooth_surface():
    from oracle.tsdf_ref import TSDFRef
    ref = TSDFRef(0.01, 0.04, res=8, stride=4)
    H, W, K = 48, 64, (60.0, 60.0, 32.0, 24.0)
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (0.5 + 0.05 * np.sin(u / 9.0) * np.cos(v / 7.0)).astype(np.float32)
    ref.integrate(depth, np.full((H, W, 3), 128, np.uint8), K, np.eye(4))
    V, C, T = ref.extract_triangle_mesh()
    assert V.shape[0] > 1000 and T.shape[0] > 2000 and np.allclose(C, 128 / 255.0, atol=1e-6)
    de = Counter()
    for a, b, c in T:
        for e in ((a, b), (b, c), (c, a)):
            de[e] += 1
    assert max(de.values()) == 1                                # manifold and consistently wound: no directed edge twice
    n = np.cross(V[T[:, 1]] - V[T[:, 0]], V[T[:, 2]] - V[T[:, 0]])
    assert (n[:, 2] < 0).all()                                  # towards the camera at the origin: from tsdf < 0 (behind) to tsdf > 0
    # the vertices lie on the measured surface (to a voxel): z of the depth map at their pixel
    uu, vv = V[:, 0] / V[:, 2] * K[0] + K[2], V[:, 1] / V[:, 2] * K[1] + K[3]
    ok = (uu > 1) & (uu < W - 2) & (vv > 1) & (vv < H - 2)
    zs = depth[np.clip(np.rint(vv[ok]).astype(int), 0, H - 1), np.clip(np.rint(uu[ok]).astype(int), 0, W - 1)]
    assert np.abs(V[ok, 2] - zs).max() < 0.012
