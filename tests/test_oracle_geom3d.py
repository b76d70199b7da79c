"""Pins oracle/geom3d_ref.py and oracle/backproject_ref.c against vectors produced by the
reference's own functions (oracle/make_golden.py; slam_utils.py:71-122, scaling_system.py:72-77)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import geom3d_ref as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chain(golden_dir):
    return np.load(os.path.join(golden_dir, "geom3d_chain.npz"))


@pytest.fixture(scope="module")
def bp(golden_dir):
    return np.load(os.path.join(golden_dir, "geom3d_backproject.npz"))


@pytest.fixture(scope="module")
def clib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libbackproject_ref.so"))
    lib.bsref_backproject.restype = ctypes.c_int64
    return lib


def test_chain_matches_reference(chain):
    g = G.pose_chain(chain["t_rel"])
    assert g.dtype == np.float64 and g.shape == (1001, 4, 4)
    # same numpy/LAPACK in the same image: bit-exact
    assert np.array_equal(g, chain["g_abs"])


def test_chain_is_float64_and_so3(chain):
    g = G.pose_chain(chain["t_rel"][:50])
    for m in g:
        r = m[:3, :3]
        assert np.allclose(r @ r.T, np.eye(3), atol=1e-12)
        assert abs(np.linalg.det(r) - 1) < 1e-12
        assert np.array_equal(m[3], [0, 0, 0, 1])


def test_inverse_list(chain):
    g = G.pose_chain(chain["t_rel"][:20])
    inv = np.stack([np.linalg.inv(m) for m in g])
    assert np.array_equal(inv, chain["inv_abs"][:21])


def test_ensure_so3(chain):
    out = np.stack([G.ensure_so3_v2(m) for m in chain["so3_in"]])
    assert np.array_equal(out, chain["so3_out"])
    assert np.linalg.det(out[0]) > 0  # reflection input is corrected to a proper rotation


def test_backproject_numpy(bp):
    xyz, idx = G.backproject(bp["depth"], tuple(bp["K"]))
    assert np.array_equal(idx, bp["idx"])
    assert np.array_equal(xyz, bp["xyz"].astype(np.float32))


def test_backproject_edge_values(bp):
    d = bp["depth"]
    _, idx = G.backproject(d, tuple(bp["K"]))
    W = d.shape[1]
    s = set(idx.tolist())
    assert 0 in s                      # 2999 -> valid
    assert 1 not in s                  # 3000 -> truncated
    assert (d.size - 1) in s           # 1 -> valid
    assert (d.size - 2) not in s       # 65535 -> truncated
    assert all(d.ravel()[i] > 0 for i in idx)


def test_backproject_c(bp, clib):
    d = np.ascontiguousarray(bp["depth"])
    H, W = d.shape
    K = np.ascontiguousarray(bp["K"], dtype=np.float64)
    xyz = np.zeros((H * W, 3), np.float32)
    idx = np.zeros(H * W, np.int32)
    m = clib.bsref_backproject(d.ctypes.data_as(ctypes.c_void_p), H, W, K.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_double(1000.0), ctypes.c_double(3.0), None,
                               xyz.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p))
    assert m == len(bp["idx"])
    assert np.array_equal(idx[:m], bp["idx"])
    assert np.array_equal(xyz[:m], bp["xyz"].astype(np.float32))


def test_backproject_c_with_pose_and_empty(bp, clib, chain):
    d = np.ascontiguousarray(bp["depth"])
    H, W = d.shape
    K = np.ascontiguousarray(bp["K"], dtype=np.float64)
    pose = np.ascontiguousarray(chain["g_abs"][37])
    xyz = np.zeros((H * W, 3), np.float32)
    idx = np.zeros(H * W, np.int32)
    m = clib.bsref_backproject(d.ctypes.data_as(ctypes.c_void_p), H, W, K.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_double(1000.0), ctypes.c_double(3.0), pose.ctypes.data_as(ctypes.c_void_p),
                               xyz.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p))
    ref_xyz, ref_idx = G.backproject(d, tuple(K), pose=pose)
    assert np.array_equal(idx[:m], ref_idx)
    assert np.allclose(xyz[:m], ref_xyz, rtol=0, atol=1e-6)
    z = np.zeros((4, 8), np.uint16)
    m0 = clib.bsref_backproject(z.ctypes.data_as(ctypes.c_void_p), 4, 8, K.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_double(1000.0), ctypes.c_double(3.0), None, None, None)
    assert m0 == 0
    assert G.backproject(z)[1].size == 0


def test_pixel_to_3d_formula():
    p = G.pixel_to_3d(300, 200, 1.5, *G.REF_INTRINSICS)
    assert np.allclose(p, [(300 - 276.4727783203125) * 1.5 / 383.1901395, (200 - 124.3335933685303) * 1.5 / 383.1901395, 1.5])


def test_kitti_pose_writer_matches_reference_file(golden_dir, tmp_path):
    """save_poses_as_kitti (drop-in for UTILS/io_utils.py:264-278): byte-identical to the file the reference's own writer
    produced for the same poses (tests/golden/kitti_poses_24.txt, made by oracle/make_golden.py)."""
    import os
    from bodyslam_amd.slam_utils import save_poses_as_kitti
    g = np.load(os.path.join(golden_dir, "geom3d_chain.npz"))
    out = tmp_path / "poses.txt"
    save_poses_as_kitti(list(g["g_abs"][:24]), str(out))
    assert out.read_text() == open(os.path.join(golden_dir, "kitti_poses_24.txt")).read()
