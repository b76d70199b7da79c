"""TSDF oracle (oracle/tsdf_ref.py): properties of the restated Open3D algorithm on analytic scenes (CPU).  Parity against Open3D
itself is unpinned (not installed); the GPU product is compared with this oracle in tests/test_tsdf_gpu.py."""
import numpy as np

from oracle.tsdf_ref import TSDFRef

K = (60.0, 60.0, 32.0, 24.0)
H, W = 48, 64


def plane(z=0.5):
    return np.full((H, W), z, dtype=np.float32)


def test_plane_surface_and_weights():
    t = TSDFRef(voxel_length=0.01, sdf_trunc=0.04, res=8, stride=4)
    color = np.zeros((H, W, 3), np.uint8)
    color[..., 0], color[..., 1] = 200, 50
    t.integrate(plane(), color, K, np.eye(4))
    assert len(t.units) > 0
    # units lie within sdf_trunc of the surface (+ one unit of slack for the box test)
    for (ix, iy, iz) in t.units:
        assert -0.04 - 0.08 <= (iz + 0.5) * 0.08 - 0.5 <= 0.04 + 0.08
    pts, cols = t.extract_point_cloud()
    assert pts.shape[0] > 100
    n = np.array([t.normal_at(p) for p in pts[:: max(1, pts.shape[0] // 60)].astype(np.float64)])
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-9) and (n[:, 2] < -0.95).all()     # a plane facing the camera at z < plane
    # the zero crossing of the projective distance of a fronto-parallel plane is the plane itself: (d - z) * mult = 0 at z = d
    assert np.abs(pts[:, 2] - 0.5).max() < 1e-3
    assert np.allclose(cols, np.array([200, 50, 0]) / 255.0, atol=1e-6)
    w1 = {k: v[..., 1].copy() for k, v in t.units.items()}
    f1 = {k: v[..., 0].copy() for k, v in t.units.items()}
    t.integrate(plane(), color, K, np.eye(4))
    for k in w1:           # the same observation again: weights double, the running mean does not move
        seen = w1[k] > 0
        assert np.array_equal(t.units[k][..., 1][seen], 2 * w1[k][seen])
        assert np.allclose(t.units[k][..., 0], f1[k], atol=1e-6)


def test_truncation_and_invalid_depth():
    t = TSDFRef(voxel_length=0.01, sdf_trunc=0.04, res=8, stride=4)
    d = plane()
    d[:, : W // 2] = 0.0            # no measurement on the left half
    t.integrate(d, None, K, np.eye(4))
    for key, v in t.units.items():
        assert v[..., 0].max() <= 1.0 and v[..., 0].min() > -1.0 - 1e-6
    pts, _ = t.extract_point_cloud()
    assert pts.shape[0] > 0 and (pts[:, 0] > -0.02).all()          # nothing is reconstructed where nothing was measured


def test_pose_moves_the_surface():
    t = TSDFRef(voxel_length=0.01, sdf_trunc=0.04, res=8, stride=4)
    pose = np.eye(4)
    pose[:3, 3] = (0.1, -0.05, 0.2)                                  # camera placed in the world
    t.integrate(plane(), None, K, np.linalg.inv(pose))
    pts, _ = t.extract_point_cloud()
    assert np.abs(pts[:, 2] - 0.7).max() < 1e-3
