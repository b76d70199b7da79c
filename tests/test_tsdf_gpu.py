"""TSDF map on the GPU (bodyslam_amd/tsdf.py + csrc/tsdf.hip) against the numpy oracle (oracle/tsdf_ref.py): same units opened,
same voxels, same extracted surface points.  Reference: BodySLAM_not_refactored/3DM/tsdf.py:5-52 (Open3D ScalableTSDFVolume;
parity against Open3D itself is unpinned, see the oracle's header)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W = 48, 64
K = (60.0, 60.0, 32.0, 24.0)


def scene(seed):
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (0.5 + 0.05 * np.sin(u / 9.0 + seed) * np.cos(v / 7.0)).astype(np.float32)
    depth[rng.random((H, W)) < 0.03] = 0.0                                  # holes
    color = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    a = 0.05 * seed
    pose = np.eye(4)
    pose[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    pose[:3, 3] = (0.03 * seed, -0.02 * seed, 0.01 * seed)
    return depth, color, np.linalg.inv(pose)


def sort_rows(p, c):
    o = np.lexsort((p[:, 2], p[:, 1], p[:, 0]))
    return p[o], c[o]


@pytest.mark.parametrize("res,stride", [(8, 4), (4, 8)])
def test_tsdf_matches_oracle(res, stride, tmp_path):
    from bodyslam_amd.tsdf import TSDF, PinholeCameraIntrinsic, RGBDImage
    from oracle.tsdf_ref import TSDFRef
    vl, trunc = 0.01, 0.04
    prod = TSDF(vl, trunc, volume_unit_resolution=res, depth_sampling_stride=stride, slab_bytes=1 << 16)   # several slabs
    ref = TSDFRef(vl, trunc, res=res, stride=stride)
    intr = PinholeCameraIntrinsic(W, H, *K)
    for seed in range(3):
        depth, color, E = scene(seed)
        prod.build_3D_map(RGBDImage(color, depth), intr, torch.from_numpy(E) if seed == 1 else E)
        ref.integrate(depth, color, K, E)
    assert set(prod.index) == set(ref.units)
    worst = 0.0
    for key, vox in ref.units.items():
        got = prod.unit(key)
        assert np.array_equal(got[..., 1], vox[..., 1]), f"weights of unit {key}"
        worst = max(worst, float(np.abs(got - vox).max()))
    assert worst < 2e-4                   # colours are 0..255 running means: a last-bit difference is 1.5e-5
    pcd = prod.extract_pcd()
    rp, rc, rn = ref.extract_point_cloud(normals=True)
    assert pcd.points.shape == rp.shape and rp.shape[0] > 500
    og, orf = np.lexsort((pcd.points[:, 2], pcd.points[:, 1], pcd.points[:, 0])), np.lexsort((rp[:, 2], rp[:, 1], rp[:, 0]))
    gp, gc, gn = pcd.points[og], pcd.colors[og], pcd.normals[og]
    rp, rc, rn = rp[orf], rc[orf], rn[orf]
    assert np.abs(gp - rp).max() < 1e-6 and np.abs(gc - rc).max() < 1e-5
    # normals (GetNormalAt): the product evaluates them at the fp64 point, the oracle at the fp32-rounded one -> 1e-4 on unit vectors
    assert np.abs(gn - rn).max() < 2e-4
    ln = np.linalg.norm(gn, axis=1)
    assert np.all((np.abs(ln - 1.0) < 1e-5) | (ln == 0.0)) and (ln > 0).mean() > 0.99
    assert (gn[:, 2] < 0).mean() > 0.9           # the tsdf grows towards the camera (at z ~ 0): normals point back at it
    # the surface is where it was put: camera-frame depth of every point of the first view's neighbourhood ~ the scene's range
    assert 0.40 < gp[:, 2].min() and gp[:, 2].max() < 0.75
    path = tmp_path / "map.ply"
    prod.save_pcd(str(path))
    raw = path.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert f"element vertex {gp.shape[0]}".encode() in head and b"property float nx" in head and len(body) == gp.shape[0] * 27
    # extract_mesh / save_mesh (tsdf.py:42-52): marching cubes on the device against the oracle's table-free restatement
    mesh = prod.extract_mesh()
    rv, rcol, rt = ref.extract_triangle_mesh()
    assert mesh.triangles.shape[0] == rt.shape[0] > 1000 and mesh.vertices.shape == rv.shape
    og, orf = np.lexsort((mesh.vertices[:, 2], mesh.vertices[:, 1], mesh.vertices[:, 0])), np.lexsort((rv[:, 2], rv[:, 1], rv[:, 0]))
    assert np.abs(mesh.vertices[og] - rv[orf]).max() < 1e-6 and np.abs(mesh.vertex_colors[og] - rcol[orf]).max() < 1e-5
    tri = mesh.triangles
    assert tri.min() == 0 and tri.max() == mesh.vertices.shape[0] - 1 and (tri[:, 0] != tri[:, 1]).all()

    def open_edges(t):
        d = {}
        for a, b, c in t:
            for e in ((a, b), (b, c), (c, a)):
                d[e] = d.get(e, 0) + 1
        assert max(d.values()) == 1                              # consistent winding
        return sum(1 for (a, b) in d if (b, a) not in d)
    assert open_edges(tri) == open_edges(rt)                      # the same rim, nothing torn inside
    mpath = tmp_path / "m.ply"
    prod.save_mesh(str(mpath))
    mh, mb = mpath.read_bytes().split(b"end_header\n", 1)
    assert f"element vertex {rv.shape[0]}".encode() in mh and f"element face {rt.shape[0]}".encode() in mh
    assert len(mb) == rv.shape[0] * 15 + rt.shape[0] * 13


def test_tsdf_copy_and_reference_parameters():
    """build_copy_3D_map leaves the original untouched; the reference's own parameters (1 mm voxels, 0.1 m truncation, 32^3 units,
    stride 8) run on a small image: a point opens the ~7^3 units around it."""
    from bodyslam_amd.tsdf import TSDF, PinholeCameraIntrinsic, RGBDImage
    depth = np.zeros((16, 16), np.float32)
    depth[8, 8] = 0.3
    intr = PinholeCameraIntrinsic(16, 16, 20.0, 20.0, 8.0, 8.0)
    t = TSDF()                                     # tsdf.py:6 defaults
    t.build_3D_map(RGBDImage(None, depth), intr, np.eye(4))
    n = len(t.index)
    assert 6 ** 3 <= n <= 8 ** 3
    t2 = t.build_copy_3D_map(RGBDImage(None, depth), intr, np.eye(4))
    # only the voxels that project onto the one measured pixel are updated (most opened units stay empty): the unit that holds
    # the surface point (0, 0, 0.3) is one of them
    L_ = 0.001 * 32
    k = (int(np.floor(0.0 / L_)), int(np.floor(0.0 / L_)), int(np.floor(0.3 / L_)))
    assert k in t.index and t.unit(k)[..., 1].max() == 1.0
    for key in t.index[:: max(1, n // 40)] + [k]:
        assert np.array_equal(t2.unit(key)[..., 1], 2 * t.unit(key)[..., 1])
    assert len(t2.index) == n
    pcd = t.extract_pcd()
    assert pcd.points.shape[0] > 0 and np.abs(pcd.points[:, 2] - 0.3).max() < 2e-3


def test_pipeline_integrates_its_frames():
    """the loop's map step (3DM/slam.py:179): run_sequence's depth + poses go into the TSDF"""
    import dataclasses
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.tsdf import TSDF
    from bodyslam_amd.zoedepth import ZoeConfig
    from oracle import cyclepose_ref as CP
    from oracle import zoedepth_ref as Z
    cfg_o = Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    cfg_p = ZoeConfig(**{k: v for k, v in dataclasses.asdict(cfg_o).items() if k in names})
    frames = make_sequence(3, 160, 192, seed=5)
    pipe = BodySlamPipeline(Z.synth_weights(cfg_o, seed=2), CP.synth_weights(seed=2), cfg_p, batch=2, target_hw=(64, 96))
    res = pipe.run_sequence(frames)
    t = TSDF(voxel_length=0.02, sdf_trunc=0.06, volume_unit_resolution=8, depth_sampling_stride=8)
    pipe.integrate_tsdf(t, frames, res)
    assert len(t.index) > 0
    w = sum(float(t.unit(k)[..., 1].sum()) for k in t.index[:50])
    assert w > 0


def test_run_slam_loop_orders_the_steps_like_the_reference():
    """SLAM._sequential_loop (3DM/slam.py:131-205): depth, MPEM, VO fusion pair by pair, chain, map.  Structural checks on a small
    configuration: the fused relatives keep MPEM's rotations, their translations are the UKF states of a VO run by hand on the
    same inputs, the chain is the chain of the fused relatives, and every frame went into the TSDF."""
    import dataclasses
    from bodyslam_amd import geom3d
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.tsdf import TSDF, create_rgbd_from_color_and_depth
    from bodyslam_amd.visual_odometry import VO
    from bodyslam_amd.zoedepth import ZoeConfig
    from oracle import cyclepose_ref as CP
    from oracle import geom3d_ref as G
    from oracle import zoedepth_ref as Z
    cfg_o = Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    cfg_p = ZoeConfig(**{k: v for k, v in dataclasses.asdict(cfg_o).items() if k in names})
    frames = make_sequence(4, 160, 192, seed=5)
    pipe = BodySlamPipeline(Z.synth_weights(cfg_o, seed=2), CP.synth_weights(seed=2), cfg_p, batch=2, target_hw=(64, 96))
    plain = pipe.run_sequence(frames)
    t = TSDF(voxel_length=0.02, sdf_trunc=0.06, volume_unit_resolution=8, depth_sampling_stride=8)
    res = pipe.run_slam_loop(frames, vo=True, tsdf=t)
    assert torch.equal(res.depth_u16, plain.depth_u16)
    tp, tf = plain.t_rel.cpu().numpy(), res.t_rel.cpu().numpy()
    assert np.array_equal(tf[:, :3, :3], tp[:, :3, :3]) and not np.allclose(tf[:, :3, 3], tp[:, :3, 3])

    class Stored:
        def __init__(self):
            self.i = 0

        def infer_relative_pose_between(self, a, b):
            return tp[self.i]

    mp = Stored()
    vo = VO(mp, intrinsic=tuple(pipe.K))
    du = res.depth_u16.cpu().numpy().view(np.uint16)
    rg = [create_rgbd_from_color_and_depth(frames[i], du[i], pipe.depth_scale, pipe.depth_trunc) for i in range(4)]
    for i in range(1, 4):
        mp.i = i - 1
        T = vo.estimate_relative_pose_between(i - 1, i, rg[i - 1], rg[i], i)
        assert np.array_equal(T.astype(np.float32), tf[i - 1])
    assert np.abs(res.g_abs.cpu().numpy() - G.pose_chain(tf)).max() < 1e-10
    assert t.n_units > 0 and t.frames_integrated == 4


def test_tsdf_streamed_and_culled_equal_the_synchronous_path():
    """build_3D_map(sync=False) + reserve_ahead + sync(): the same units and voxels, bit for bit, as one round trip per frame; and the
    block culling of the integrate kernel (bounding sphere outside the frustum) changes nothing"""
    import os
    from bodyslam_amd.tsdf import TSDF, PinholeCameraIntrinsic, RGBDImage
    vl, trunc = 0.01, 0.04
    intr = PinholeCameraIntrinsic(W, H, *K)

    def run(streamed, cull=True):
        if not cull:
            os.environ["BS_TSDF_NO_CULL"] = "1"
        try:
            t = TSDF(vl, trunc, volume_unit_resolution=8, depth_sampling_stride=4, slab_bytes=1 << 16)
            if streamed:
                t.reserve_ahead(4)
            for seed in range(4):
                depth, color, E = scene(seed)
                t.build_3D_map(RGBDImage(color, depth), intr, E, sync=not streamed)
            if streamed:
                n, _ = t.sync()
                assert n == t.n_units > 0
            return t
        finally:
            os.environ.pop("BS_TSDF_NO_CULL", None)

    a, b, c = run(False), run(True), run(False, cull=False)
    assert set(a.index) == set(b.index) == set(c.index)
    for key in a.index:
        assert np.array_equal(a.unit(key), b.unit(key)) and np.array_equal(a.unit(key), c.unit(key)), key
    # capacity: a stream that runs out of blocks reports it at the sync (the flag is sticky), it does not corrupt memory
    t = TSDF(vl, trunc, volume_unit_resolution=8, depth_sampling_stride=4, slab_bytes=1 << 16, max_units=64)
    depth, color, E = scene(0)
    t.build_3D_map(RGBDImage(color, depth), intr, E, sync=False)
    t.build_3D_map(RGBDImage(color, depth), intr, E, sync=False)
    with pytest.raises(Exception, match="blocks"):
        t.sync()


def test_tsdf_frame_batch_equals_frame_by_frame():
    """build_3D_map_batch: a run of frames integrated in one pass over the map (every touched voxel loaded once, its frames applied in
    order in registers) leaves the same units and the same voxels, bit for bit, as build_3D_map frame by frame -- with colours and
    without, across a chunk border (65 frames > the 64-frame mask), into a map that already holds frames, and with culling off"""
    import os
    from bodyslam_amd.tsdf import TSDF, PinholeCameraIntrinsic, RGBDImage
    vl, trunc = 0.01, 0.04
    intr = PinholeCameraIntrinsic(W, H, *K)
    scenes = [scene(seed) for seed in range(5)]

    def frames(n, with_color):
        return [(RGBDImage(scenes[i % 5][1] if with_color else None, scenes[i % 5][0]), scenes[i % 5][2]) for i in range(n)]

    def check(n, with_color, pre=0, cull=True):
        if not cull:
            os.environ["BS_TSDF_NO_CULL"] = "1"
        try:
            a = TSDF(vl, trunc, volume_unit_resolution=8, depth_sampling_stride=4, slab_bytes=1 << 16)
            b = TSDF(vl, trunc, volume_unit_resolution=8, depth_sampling_stride=4, slab_bytes=1 << 16)
            fr = frames(pre + n, with_color)
            for r, E in fr:
                a.build_3D_map(r, intr, E)
            for r, E in fr[:pre]:
                b.build_3D_map(r, intr, E)
            b.build_3D_map_batch([r for r, _ in fr[pre:]], intr, [E for _, E in fr[pre:]])
            nb, _ = b.sync()
            assert nb == a.n_units > 0 and b.frames_integrated == a.frames_integrated == pre + n
            assert set(a.index) == set(b.index)
            for key in a.index:
                assert np.array_equal(a.unit(key), b.unit(key)), (n, with_color, pre, key)
            assert int(b.table_fmask.abs().sum()) == 0                 # the discovery masks are clean for the next batch
            return a, b
        finally:
            os.environ.pop("BS_TSDF_NO_CULL", None)

    check(4, True)
    check(3, False, pre=2)
    check(65, True)
    a, b = check(4, True, cull=False)
    pa, pb = a.extract_pcd(), b.extract_pcd()
    assert pa.points.shape == pb.points.shape and np.array_equal(np.sort(pa.points.view("f4,f4,f4"), axis=0), np.sort(pb.points.view("f4,f4,f4"), axis=0))
    # a full map is reported, not overrun
    # ... and by the batch call itself, before anything of the batch is integrated: a direct caller that never calls sync() must not
    # lose frames silently (round-3 advisor), and the discovery masks must be clean again for the next batch
    t = TSDF(vl, trunc, volume_unit_resolution=8, depth_sampling_stride=4, slab_bytes=1 << 16, max_units=64)
    fr = frames(2, True)
    with pytest.raises(Exception, match="max_units|unit table full"):
        t.build_3D_map_batch([r for r, _ in fr], intr, [E for _, E in fr])
    assert t.frames_integrated == 0 and int(t.table_fmask.abs().sum()) == 0 and int(t.counters[2]) == 0


def test_slam_loop_reference_order_640x480():
    """run_slam_loop at the bench's frame size with the reference's own TSDF parameters (1 mm voxels, 0.1 m truncation, 32^3 units, stride
    8), VO fusion on, pose graph every 2 frames: the per-frame order of SLAM._sequential_loop (3DM/slam.py:131-205) -- frames 2 and 4 take
    the optimise branch, the poses do not move (odometry edges only), so those frames are NOT integrated (slam.py:159-179) -- checked
    against the numpy TSDF oracle fed with the loop's own depth maps and poses."""
    import dataclasses
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.tsdf import TSDF, create_rgbd_from_color_and_depth
    from bodyslam_amd.zoedepth import ZoeConfig
    from oracle import cyclepose_ref as CP
    from oracle import geom3d_ref as G
    from oracle import zoedepth_ref as Z
    from oracle.tsdf_ref import TSDFRef
    cfg_o = Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    cfg_p = ZoeConfig(**{k: v for k, v in dataclasses.asdict(cfg_o).items() if k in names})
    N = 5
    frames = make_sequence(N, 480, 640, seed=4)
    pipe = BodySlamPipeline(Z.synth_weights(cfg_o, seed=2), CP.synth_weights(seed=2), cfg_p, batch=2)
    t = TSDF()                                                  # the reference's parameters (tsdf.py:6-12)
    seen = []
    res = pipe.run_slam_loop(frames, vo=True, tsdf=t, posegraph_every=2, on_frame=lambda i, pose, pcd: seen.append(i))
    assert seen == list(range(N)) and pipe.last_tsdf is t       # nothing moved: no rebuild
    g = res.g_abs.cpu().numpy()
    tf = res.t_rel.cpu().numpy()
    assert np.abs(g - G.pose_chain(tf)).max() < 1e-10           # the chain of the fused relatives
    du = res.depth_u16.cpu().numpy().view(np.uint16)
    ref = TSDFRef(0.001, 0.1, res=32, stride=8)
    integrated = [0, 1, 3]
    for i in integrated:
        rg = create_rgbd_from_color_and_depth(frames[i], du[i], pipe.depth_scale, pipe.depth_trunc)
        ref.integrate(rg.depth, rg.color, tuple(pipe.K), g[i])
    assert t.frames_integrated == len(integrated)
    assert set(t.index) == set(ref.units)
    keys = sorted(ref.units, key=lambda k: -float(ref.units[k][..., 1].sum()))[:12] + list(ref.units)[:: max(1, len(ref.units) // 12)]
    for key in keys:
        got, want = t.unit(key), ref.units[key]
        assert np.array_equal(got[..., 1], want[..., 1]), key
        assert np.abs(got - want).max() < 2e-4, key
    assert 2.0 <= max(float(v[..., 1].max()) for v in ref.units.values()) <= float(len(integrated))
    pcd = t.extract_pcd()
    assert pcd.points.shape[0] > 1000


def test_run_slam_loop_edge_cases():
    """one frame, two frames, no VO, no map, a point cloud per frame (slam.py:195), points kept: the loop's options on a small configuration"""
    import dataclasses
    from bodyslam_amd.pipeline import BodySlamPipeline
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.tsdf import TSDF
    from bodyslam_amd.zoedepth import ZoeConfig
    from oracle import cyclepose_ref as CP
    from oracle import zoedepth_ref as Z
    cfg_o = Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    cfg_p = ZoeConfig(**{k: v for k, v in dataclasses.asdict(cfg_o).items() if k in names})
    frames = make_sequence(5, 160, 192, seed=6)
    pipe = BodySlamPipeline(Z.synth_weights(cfg_o, seed=2), CP.synth_weights(seed=2), cfg_p, batch=2, target_hw=(64, 96))

    def volume():
        return TSDF(voxel_length=0.02, sdf_trunc=0.06, volume_unit_resolution=8, depth_sampling_stride=8)

    whole = pipe.run_slam_loop(frames, vo=True, tsdf=volume(), keep_points=True)
    assert whole.g_abs.shape == (5, 4, 4) and whole.t_rel.shape == (4, 4, 4) and len(whole.points) == 5
    c = whole.point_counts.cpu().tolist()
    assert all(whole.points[i][0].shape == (c[i], 3) and c[i] > 0 for i in range(5))
    # one frame: the identity pose, nothing to chain, the frame is in the map
    t1 = volume()
    one = pipe.run_slam_loop(frames[:1], vo=True, tsdf=t1)
    assert one.t_rel.shape[0] == 0 and np.array_equal(one.g_abs[0].cpu().numpy(), np.eye(4)) and t1.frames_integrated == 1 and t1.n_units > 0
    assert torch.equal(one.depth_u16[0], whole.depth_u16[0])
    # two frames = the first pair of the long run
    two = pipe.run_slam_loop(frames[:2], vo=True, tsdf=volume())
    assert np.array_equal(two.t_rel.cpu().numpy(), whole.t_rel[:1].cpu().numpy()) and torch.equal(two.g_abs, whole.g_abs[:2])
    # no VO: the relatives are MPEM's own; no map: nothing else changes
    plain = pipe.run_sequence(frames)
    novo = pipe.run_slam_loop(frames, vo=False, tsdf=None)
    assert np.array_equal(novo.t_rel.cpu().numpy(), plain.t_rel.cpu().numpy()) and np.abs((novo.g_abs - plain.g_abs).cpu().numpy()).max() < 1e-12
    assert torch.equal(novo.depth_u16, whole.depth_u16) and pipe.last_tsdf is None
    # a point cloud per frame, as the reference extracts it: it grows with the map and ends as the batched run's
    clouds = []
    t2 = volume()
    per = pipe.run_slam_loop(frames, vo=True, tsdf=t2, extract_every_frame=True, on_frame=lambda i, pose, pcd: clouds.append(pcd.points.shape[0]))
    assert len(clouds) == 5 and clouds[0] > 0 and clouds[-1] >= clouds[0] and torch.equal(per.g_abs, whole.g_abs)
    ref_map = volume()
    pipe.run_slam_loop(frames, vo=True, tsdf=ref_map)
    assert t2.extract_pcd().points.shape[0] == ref_map.extract_pcd().points.shape[0] == clouds[-1]


def test_empty_observations_do_nothing():
    """frames without a single valid depth: the map stays empty (frame by frame and batched), the odometry leaves the identity"""
    from bodyslam_amd.rgbd_odometry import RGBDOdometry
    from bodyslam_amd.tsdf import TSDF, PinholeCameraIntrinsic, RGBDImage
    intr = PinholeCameraIntrinsic(W, H, *K)
    depth0 = np.zeros((H, W), np.float32)
    color = np.zeros((H, W, 3), np.uint8)
    t = TSDF(0.01, 0.04, volume_unit_resolution=8, depth_sampling_stride=4, slab_bytes=1 << 16)
    t.build_3D_map(RGBDImage(color, depth0), intr, np.eye(4))
    t.build_3D_map_batch([RGBDImage(color, depth0)] * 3, intr, [np.eye(4)] * 3)
    n, touched = t.sync()
    assert n == 0 and touched == 0 and t.frames_integrated == 4
    assert t.extract_pcd().points.shape == (0, 3) and t.extract_mesh().triangles.shape == (0, 3)
    # one real frame after the empty ones still lands
    d, c, E = scene(0)
    t.build_3D_map_batch([RGBDImage(color, depth0), RGBDImage(c, d)], intr, [np.eye(4), E])
    assert t.sync()[0] > 0 and t.extract_pcd().points.shape[0] > 0
    odo = RGBDOdometry(K)
    cols = torch.zeros(3, H, W, 3, dtype=torch.uint8, device="cuda")
    deps = torch.zeros(3, H, W, device="cuda")
    T = odo.track_block(cols, deps).cpu().numpy().reshape(-1, 3, 4)
    assert T.shape[0] == 2 and np.array_equal(T[0], np.eye(4)[:3]) and np.array_equal(T[1], np.eye(4)[:3])
