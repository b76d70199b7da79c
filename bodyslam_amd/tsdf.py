"""TSDF map of the SLAM loop: drop-in for BodySLAM_not_refactored/3DM/tsdf.py:5-52.

The reference's ``TSDF`` wraps Open3D's ``ScalableTSDFVolume(voxel_length=0.001, sdf_trunc=0.1, RGB8, volume_unit_resolution=32,
depth_sampling_stride=8)``: ``build_3D_map(rgbd, intrinsic, extrinsic)`` integrates a frame (3DM/slam.py:117,179 -- it passes the
accumulated pose as ``extrinsic``), ``extract_pcd`` / ``save_pcd`` read the surface points back (:126,191,195).  Same class, method
names and defaults here.  The voxels live in HBM (one contiguous block of res^3 x 5 fp32 per volume unit, allocated in slabs and
zero-filled once) and are integrated / extracted by the HIP kernels of csrc/tsdf.hip; which units exist is an open-addressing hash
table in HBM (Open3D: an unordered_map on the host), filled per frame from the strided depth sample exactly as Open3D does.  With the reference's parameters a
unit is 3.2 cm wide and a point opens the ~7^3 units within 0.1 m of it; a 640x480 frame of a surface at 12 cm touches ~1 600 units
= 1 GB of voxel state and takes ~1 ms (measured, tools/probes/tsdf_full_size.py: unit discovery 0.1 ms, integrate kernel 0.27 ms),
metre-scale scenes proportionally more: the voxel store is sized for 288 GB of HBM.  Open3D is not vendored and not installable offline: parity against it is unpinned (oracle/tsdf_ref.py
restates the same algorithm in numpy; tests/ compare the two).  ``extract_pcd`` returns points, colours and normals, as Open3D's
``extract_point_cloud``; ``extract_mesh`` / ``save_mesh`` (tsdf.py:42-52, called on the last frame at 3DM/slam.py:189-193) run marching
cubes over every voxel cube on the device (bs_tsdf_mesh) with the case table of bodyslam_amd/marching_cubes.py and merge the
vertices of shared cube edges on the host.
"""
from __future__ import annotations

import copy
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib as L


@dataclass
class PinholeCameraIntrinsic:
    """o3d.camera.PinholeCameraIntrinsic(width, height, fx, fy, cx, cy)"""
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float

    @property
    def intrinsic_matrix(self) -> np.ndarray:
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])


@dataclass
class RGBDImage:
    """o3d.geometry.RGBDImage: colour u8 [H, W, 3] and depth fp32 [H, W] in metres (0 = no measurement)"""
    color: Optional[np.ndarray]
    depth: np.ndarray


def create_rgbd_from_color_and_depth(color_u8, depth_u16, depth_scale: float = 1000.0, depth_trunc: float = 3.0) -> RGBDImage:
    """o3d.geometry.RGBDImage.create_from_color_and_depth(..., convert_rgb_to_intensity=False) as the reference calls it
    (3DM/slam_utils.py:212-220): depth / depth_scale as fp32, values >= depth_trunc -> 0."""
    d = np.asarray(depth_u16).astype(np.float32) / np.float32(depth_scale)
    d[d >= np.float32(depth_trunc)] = 0.0
    return RGBDImage(None if color_u8 is None else np.ascontiguousarray(np.asarray(color_u8, dtype=np.uint8)), d)


@dataclass
class PointCloud:
    points: np.ndarray                      # [M, 3] float32
    colors: np.ndarray                      # [M, 3] float32 in [0, 1]
    normals: Optional[np.ndarray] = None    # [M, 3] float32, unit length (zero where the tsdf gradient vanishes)


@dataclass
class TriangleMesh:
    vertices: np.ndarray                    # [V, 3] float32
    vertex_colors: np.ndarray               # [V, 3] float32 in [0, 1]
    triangles: np.ndarray                   # [T, 3] int32, wound so that the normal points from tsdf < 0 to tsdf > 0


BATCH_MAX = 64              # frames per pass of build_3D_map_batch (BS_TSDF_BATCH_MAX: one bit per frame in a unit's mask)
_OFF = 1 << 20              # unit indices are packed as three 21-bit fields (csrc/tsdf.hip ts_pack)


class TSDF:
    def __init__(self, voxel_length: float = 0.001, sdf_trunc: float = 0.1, volume_unit_resolution: int = 32,
                 depth_sampling_stride: int = 8, device: int = 0, slab_bytes: int = 1 << 30, max_units: Optional[int] = None):
        """max_units bounds the map (default: as many 32^3-voxel-sized blocks as fit 64 GB); the unit table is 4x that, a power of two."""
        self.voxel_length, self.sdf_trunc = float(voxel_length), float(sdf_trunc)
        self.res, self.stride = int(volume_unit_resolution), int(depth_sampling_stride)
        self.unit_length = self.voxel_length * self.res
        self.dev = torch.device("cuda", device)
        L.init(device)
        self.unit_floats = self.res ** 3 * 5
        self.slab_units = max(1, slab_bytes // (self.unit_floats * 4))
        self.max_units = int(max_units) if max_units else max(1024, min(1 << 20, (64 << 30) // (self.unit_floats * 4)))
        self.table_cap = 256
        while self.table_cap < 4 * self.max_units:
            self.table_cap *= 2
        self.max_slabs = -(-self.max_units // self.slab_units)
        self._alloc_state()

    def _alloc_state(self):
        d = self.dev
        self.slabs = []                                                   # fp32 [slab_units, unit_floats] tensors, zero-filled
        self.slab_base = torch.zeros(self.max_slabs, dtype=torch.int64, device=d)
        self.table_keys = torch.full((self.table_cap,), -1, dtype=torch.int64, device=d)
        self.table_slots = torch.full((self.table_cap,), -1, dtype=torch.int32, device=d)
        self.table_stamp = torch.full((self.table_cap,), -1, dtype=torch.int32, device=d)
        self.unit_index = torch.zeros(self.max_units, 3, dtype=torch.int32, device=d)
        self.counters = torch.zeros(3, dtype=torch.int32, device=d)
        self.touched = torch.zeros(self.max_units, dtype=torch.int32, device=d)
        self.n_units, self.frame_id = 0, 0                                # frame_id: discovery passes so far (the stamp of the unit table)
        self.frames_integrated = 0
        self._max_new_per_frame, self._frames_since_sync = 0, 0

    # ---- block capacity ------------------------------------------------------------------------------
    @property
    def alloc_units(self) -> int:
        """blocks that exist (zero-filled slabs): the unit table never hands out more"""
        return min(len(self.slabs) * self.slab_units, self.max_units)

    def reserve(self, units: int) -> None:
        """make sure at least `units` blocks exist (new slabs are allocated and zero-filled here, not inside a frame)"""
        units = min(int(units), self.max_units)
        while len(self.slabs) * self.slab_units < units:
            self.slabs.append(torch.zeros(self.slab_units, self.unit_floats, device=self.dev))
            self.slab_base[len(self.slabs) - 1] = self.slabs[-1].data_ptr()

    def reserve_ahead(self, n_frames: int) -> None:
        """capacity for `n_frames` un-synchronised frames: the known unit count + 2 048 (a first view of a scene opens ~1 500 units at
        endoscopic range) + per frame four times what the frames of the last stream opened on average, at least 128 (measured on the
        bench's synthetic sequence: ~65).  A stream that still runs out is reported by sync() -- 288 GB of HBM is what makes the
        generous bound affordable (a block is 655 KB)."""
        self.reserve(self.n_units + 2048 + n_frames * max(128, 4 * self._max_new_per_frame))

    def n_units_known(self) -> int:
        return self.n_units

    def discover(self, rgbd: "RGBDImage", intrinsic: "PinholeCameraIntrinsic", extrinsic) -> None:
        """unit discovery of a frame WITHOUT integrating it (no round trip): the units it needs enter the table and take blocks as far as
        blocks exist.  ``reserve_discovered()`` then makes the missing blocks -- two cheap passes over a batch of frames replace a guess
        at how many units a stream will open."""
        self._touch_integrate(rgbd, intrinsic, extrinsic, integrate=False)

    def reserve_discovered(self, margin: int = 256) -> int:
        """after a run of discover() calls: one round trip for the unit count and the number of table entries still without a block;
        allocates what is missing (+ margin) and clears the overflow flag those frames may have raised.  Returns the blocks added."""
        n_units = int(self.counters[0])
        missing = int(((self.table_keys != -1) & (self.table_slots < 0)).sum())
        before = self.alloc_units
        self.reserve(n_units + missing + margin)
        self.counters[2] = 0
        self.n_units = n_units
        self._frames_since_sync = 0
        return self.alloc_units - before

    def sync(self):
        """the round trip a stream of build_3D_map(sync=False) calls owes: unit counts, and the overflow check"""
        n_units, n_touched, overflow = (int(v) for v in self.counters.cpu())
        if overflow:
            self.counters[2] = 0
            raise L.BodySlamHipError("TSDF: " + ("unit table full" if overflow == 1 else
                                                 f"a frame needed more than the {self.alloc_units} blocks that existed (reserve() more ahead of a "
                                                 f"stream, or raise max_units={self.max_units})"))
        self._max_new_per_frame = max(self._max_new_per_frame, (n_units - self.n_units) // max(self._frames_since_sync, 1) + 1)
        self._frames_since_sync = 0
        self.n_units, self.last_units = n_units, n_touched
        return n_units, n_touched

    # ---- the reference's surface -----------------------------------------------------------------
    def build_3D_map(self, rgbd: RGBDImage, intrinsic: PinholeCameraIntrinsic, extrinsic, sync: bool = True) -> None:
        self._touch_integrate(rgbd, intrinsic, extrinsic, sync=sync)

    def _touch_integrate(self, rgbd: RGBDImage, intrinsic: PinholeCameraIntrinsic, extrinsic, sync: bool = False, integrate: bool = True) -> None:
        """ScalableTSDFVolume.integrate(rgbd, intrinsic, extrinsic).  sync=True (one frame at a time, as the reference calls it): the
        unit count is read back (12 bytes) and slabs are added on demand.  sync=False: nothing is read back -- the frame is enqueued
        against the blocks that exist (reserve / reserve_ahead) and ``sync()`` later collects the counts and the overflow flag."""
        d_dev = (rgbd.depth if isinstance(rgbd.depth, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(rgbd.depth, dtype=np.float32))))
        d_dev = d_dev.to(device=self.dev, dtype=torch.float32).contiguous()                    # numpy, host or device tensors
        H, W = d_dev.shape
        E = np.ascontiguousarray(np.asarray(self._np(extrinsic), dtype=np.float64))
        K = np.array([intrinsic.fx, intrinsic.fy, intrinsic.cx, intrinsic.cy], dtype=np.float64)
        pose12 = np.ascontiguousarray(np.linalg.inv(E)[:3].reshape(12))
        e12 = np.ascontiguousarray(E[:3].reshape(12))
        c_dev = None
        if rgbd.color is not None:
            c_dev = rgbd.color if isinstance(rgbd.color, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(rgbd.color, dtype=np.uint8)))
            c_dev = c_dev.to(device=self.dev, dtype=torch.uint8).contiguous()
        lib, st = L.load_library(), L.stream_ptr()
        if not self.slabs:
            self.reserve(1)

        def touch():
            self.frame_id += 1
            L.check(lib.bs_tsdf_touch(L.p(d_dev), H, W, self.stride, K.ctypes.data_as(C.c_void_p), pose12.ctypes.data_as(C.c_void_p), self.unit_length,
                                      self.sdf_trunc, L.p(self.table_keys), L.p(self.table_slots), L.p(self.table_stamp), self.table_cap, self.frame_id,
                                      L.p(self.unit_index), self.alloc_units, L.p(self.counters), L.p(self.touched), st), "bs_tsdf_touch")

        integrate_too = integrate

        def integrate(n_hint):
            # (K / pose are copied into the kernel arguments at launch: nothing here has to outlive the call)
            L.check(lib.bs_tsdf_integrate(L.p(d_dev), L.p(c_dev), H, W, K.ctypes.data_as(C.c_void_p), e12.ctypes.data_as(C.c_void_p), L.p(self.unit_index),
                                          L.p(self.touched), n_hint, L.p(self.slab_base), self.slab_units, self.res, self.voxel_length, self.sdf_trunc,
                                          L.p(self.counters[1:]), st), "bs_tsdf_integrate")

        self._frames_since_sync += 1
        self.frames_integrated += int(integrate_too)
        if not sync:
            touch()
            if integrate_too:
                integrate(self.alloc_units)
            return
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        touch()
        ev[1].record()
        while True:
            n_units, n_touched, overflow = (int(v) for v in self.counters.cpu())          # the one host round trip of the step (12 bytes)
            if overflow == 2 and self.alloc_units < self.max_units:
                # more new units than blocks existed: add slabs and discover again (a fresh frame id re-stamps this frame's units)
                self.counters[2] = 0
                self.reserve(max(2 * self.alloc_units, self.alloc_units + self.slab_units))
                touch()
                continue
            break
        if overflow:
            self.counters[2] = 0
            raise L.BodySlamHipError("TSDF: " + ("unit table full" if overflow == 1 else f"more than max_units={self.max_units} volume units")
                                     + "; construct TSDF with a larger max_units")
        self._max_new_per_frame = max(self._max_new_per_frame, n_units - self.n_units)
        self._frames_since_sync = 0
        self.n_units, self.last_units = n_units, n_touched
        ev[2].record()
        integrate(max(n_touched, 1))
        ev[3].record()
        ev[3].synchronize()
        self.last_touch_ms, self.last_kernel_ms = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])      # diagnostics

    def build_3D_map_batch(self, rgbds: Sequence["RGBDImage"], intrinsic: "PinholeCameraIntrinsic", extrinsics) -> None:
        """``build_3D_map`` for a run of frames, in order, as ONE pass over the map (csrc/tsdf.hip, "a batch of frames at once"): every
        voxel the run touches is loaded once, takes its frames in ascending order in registers and is stored once.  The blocks end
        up bit for bit as the frame-by-frame calls leave them; the cost is three launches and one 12-byte round trip (the block
        reservation) per 64 frames instead of three launches per frame.  Device images (torch tensors) are used in place."""
        n = len(rgbds)
        assert len(extrinsics) == n
        if n == 0:
            return
        lib, st = L.load_library(), L.stream_ptr()
        if not self.slabs:
            self.reserve(1)
        if getattr(self, "table_fmask", None) is None:
            self.table_fmask = torch.zeros(self.table_cap, dtype=torch.int64, device=self.dev)
            self.unit_mask = torch.zeros(self.max_units, dtype=torch.int64, device=self.dev)
            self.frames_dev = torch.zeros(BATCH_MAX * 256, dtype=torch.uint8, device=self.dev)      # BS_TSDF_FRAME_BYTES per frame
        K = np.array([intrinsic.fx, intrinsic.fy, intrinsic.cx, intrinsic.cy], dtype=np.float64)
        for a in range(0, n, BATCH_MAX):
            chunk = rgbds[a:a + BATCH_MAX]
            m = len(chunk)
            depth = [self._image(r.depth, torch.float32) for r in chunk]
            has_color = chunk[0].color is not None
            assert all((r.color is not None) == has_color for r in chunk), "either every frame of a batch has a colour image or none"
            color = [self._image(r.color, torch.uint8) for r in chunk] if has_color else None
            H, W = depth[0].shape
            assert all(d.shape == (H, W) for d in depth)
            E = np.ascontiguousarray(np.stack([np.asarray(self._np(e), dtype=np.float64) for e in extrinsics[a:a + m]]).reshape(m, 16))
            P = np.ascontiguousarray(np.linalg.inv(E.reshape(m, 4, 4)).reshape(m, 16))
            dptr = np.array([d.data_ptr() for d in depth], dtype=np.uint64)
            cptr = np.array([c.data_ptr() for c in color], dtype=np.uint64) if has_color else None
            L.check(lib.bs_tsdf_frames_upload(dptr.ctypes.data_as(C.c_void_p), cptr.ctypes.data_as(C.c_void_p) if has_color else None,
                                              K.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p), P.ctypes.data_as(C.c_void_p), m,
                                              L.p(self.frames_dev), st), "bs_tsdf_frames_upload")
            L.check(lib.bs_tsdf_touch_batch(L.p(self.frames_dev), m, H, W, self.stride, self.unit_length, self.sdf_trunc, L.p(self.table_keys),
                                            L.p(self.table_fmask), self.table_cap, L.p(self.counters), st), "bs_tsdf_touch_batch")
            # the one round trip of the batch: how many blocks the discovered units need
            # (+ the overflow flag the discovery may have raised: 1 = unit table full)
            need = torch.stack([self.counters[0].to(torch.int64), ((self.table_keys != -1) & (self.table_slots < 0)).sum(),
                                self.counters[2].to(torch.int64)]).cpu()
            if int(need[2]) == 1 or int(need[0]) + int(need[1]) > self.max_units:
                # nothing of this batch has been integrated yet: clear the flag and the batch's frame bits, then refuse
                self.counters[2] = 0
                self.table_fmask.zero_()
                raise L.BodySlamHipError("TSDF: " + ("unit table full" if int(need[2]) == 1 else
                                                     f"the batch needs {int(need[0]) + int(need[1])} volume units, more than max_units={self.max_units}")
                                         + "; construct TSDF with a larger max_units")
            self.reserve(int(need[0]) + int(need[1]))
            L.check(lib.bs_tsdf_integrate_batch(L.p(self.frames_dev), m, H, W, L.p(self.table_keys), L.p(self.table_slots), L.p(self.table_fmask),
                                                self.table_cap, L.p(self.unit_index), self.alloc_units, L.p(self.counters), L.p(self.touched),
                                                L.p(self.unit_mask), L.p(self.slab_base), self.slab_units, self.res, self.voxel_length, self.sdf_trunc,
                                                st), "bs_tsdf_integrate_batch")
            self.frames_integrated += m
            self._frames_since_sync += m
            # (the assignment cannot run out of blocks after the reserve above; its flag is still collected by sync())
            # every discovered unit has a block now (unless the map is full: sync() reports that), so the count is known without
            # another round trip -- extract_pcd / extract_mesh right after a batch see the whole map
            self.n_units = min(int(need[0]) + int(need[1]), self.alloc_units)

    def _image(self, x, dtype) -> torch.Tensor:
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
        return t.to(device=self.dev, dtype=dtype).contiguous()

    def build_copy_3D_map(self, rgbd, intrinsic, extrinsic) -> "TSDF":
        other = copy.copy(self)
        for name in ("table_keys", "table_slots", "table_stamp", "unit_index", "counters", "touched"):
            setattr(other, name, getattr(self, name).clone())
        other.slabs = [s.clone() for s in self.slabs]
        other.slab_base = torch.zeros_like(self.slab_base)
        for n, sl in enumerate(other.slabs):
            other.slab_base[n] = sl.data_ptr()
        other.build_3D_map(rgbd, intrinsic, extrinsic)
        return other

    def extract_pcd(self) -> PointCloud:
        U = self.n_units
        if U == 0:
            return PointCloud(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
        count = torch.zeros(U, dtype=torch.int32, device=self.dev)
        lib = L.load_library()
        args = (L.p(self.unit_index), U, L.p(self.table_keys), L.p(self.table_slots), self.table_cap, L.p(self.slab_base), self.slab_units, self.res,
                self.voxel_length, L.p(count))
        L.check(lib.bs_tsdf_extract(*args, None, None, None, None, L.stream_ptr()), "bs_tsdf_extract")
        counts = count.cpu().numpy().astype(np.int64)
        total = int(counts.sum())
        pts = torch.empty(max(total, 1), 3, device=self.dev)
        cols = torch.empty(max(total, 1), 3, device=self.dev)
        nrm = torch.empty(max(total, 1), 3, device=self.dev)
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)).to(self.dev)
        L.check(lib.bs_tsdf_extract(*args, L.p(off), L.p(pts), L.p(cols), L.p(nrm), L.stream_ptr()), "bs_tsdf_extract")
        return PointCloud(pts[:total].cpu().numpy(), cols[:total].cpu().numpy(), nrm[:total].cpu().numpy())

    def save_pcd(self, saving_path: str) -> None:
        write_ply(saving_path, self.extract_pcd())

    def extract_mesh(self) -> TriangleMesh:
        """ScalableTSDFVolume.extract_triangle_mesh(): marching cubes over every voxel cube whose eight corners carry weight"""
        from . import marching_cubes as MC
        U = self.n_units
        if U == 0:
            return TriangleMesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
        if getattr(self, "_mc_tab", None) is None:
            tab = np.concatenate([MC.TRI_TABLE.reshape(-1), np.array(MC.EDGES, dtype=np.int32).reshape(-1)]).astype(np.int32)
            self._mc_tab = torch.from_numpy(tab).to(self.dev)
        lib = L.load_library()
        count = torch.zeros(U, dtype=torch.int32, device=self.dev)
        err = torch.zeros(1, dtype=torch.int32, device=self.dev)
        args = (L.p(self.unit_index), U, L.p(self.table_keys), L.p(self.table_slots), self.table_cap, L.p(self.slab_base), self.slab_units, self.res,
                self.voxel_length, L.p(self._mc_tab), int(MC.TRI_TABLE.shape[1]), L.p(count))
        L.check(lib.bs_tsdf_mesh(*args, None, None, None, None, L.p(err), L.stream_ptr()), "bs_tsdf_mesh")
        counts = count.cpu().numpy().astype(np.int64)
        total = int(counts.sum())
        if total == 0:
            return TriangleMesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)).to(self.dev)
        verts = torch.empty(3 * total, 3, device=self.dev)
        cols = torch.empty(3 * total, 3, device=self.dev)
        keys = torch.empty(3 * total, dtype=torch.int64, device=self.dev)
        L.check(lib.bs_tsdf_mesh(*args, L.p(off), L.p(verts), L.p(cols), L.p(keys), L.p(err), L.stream_ptr()), "bs_tsdf_mesh")
        if int(err.cpu()):
            raise L.BodySlamHipError("TSDF.extract_mesh: the map spans more than 2^20 voxels along an axis (vertex identity overflow)")
        # equal edge identities are one vertex (every cube that shares the edge computed the same position)
        # (merged on the device: a radix sort of the keys; on the host np.unique took 1 s for 17 M corners -- longer than a 256-frame loop)
        uniq, inverse = torch.unique(keys, sorted=True, return_inverse=True)
        first = torch.full((uniq.shape[0],), keys.shape[0], dtype=torch.int64, device=self.dev)
        first.scatter_reduce_(0, inverse, torch.arange(keys.shape[0], device=self.dev), reduce="amin")      # the first corner of every vertex
        return TriangleMesh(verts[first].cpu().numpy(), cols[first].cpu().numpy(), inverse.view(-1, 3).to(torch.int32).cpu().numpy())

    def save_mesh(self, saving_path: str) -> None:
        write_ply_mesh(saving_path, self.extract_mesh())

    # ---- views of the device state (tests, diagnostics) ---------------------------------------------
    @staticmethod
    def _np(m):
        return m.detach().cpu().numpy() if isinstance(m, torch.Tensor) else np.asarray(m)

    @property
    def index(self) -> list:
        """unit indices (ix, iy, iz) in block order"""
        return [tuple(int(v) for v in r) for r in self.unit_index[:self.n_units].cpu().numpy()]

    def unit(self, key) -> np.ndarray:
        """voxels of one unit as fp32 [res, res, res, 5]"""
        s = self.index.index(tuple(int(v) for v in key))
        return self.slabs[s // self.slab_units][s % self.slab_units].view(self.res, self.res, self.res, 5).cpu().numpy()


def write_ply(path: str, pcd: PointCloud) -> None:
    """binary little-endian PLY: x y z (float32) [nx ny nz (float32)] red green blue (uchar) -- the container o3d.io.write_point_cloud
    produces for a .ply path (Open3D stores coordinates and normals as doubles; the values here are fp32)"""
    n = pcd.points.shape[0]
    has_n = pcd.normals is not None
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")] + ([("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")] if has_n else []) + \
             [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.empty(n, dtype=fields)
    rec["x"], rec["y"], rec["z"] = pcd.points[:, 0], pcd.points[:, 1], pcd.points[:, 2]
    if has_n:
        rec["nx"], rec["ny"], rec["nz"] = pcd.normals[:, 0], pcd.normals[:, 1], pcd.normals[:, 2]
    c = np.clip(np.rint(pcd.colors * 255.0), 0, 255).astype(np.uint8) if n else np.zeros((0, 3), np.uint8)
    rec["red"], rec["green"], rec["blue"] = c[:, 0], c[:, 1], c[:, 2]
    props = "property float x\nproperty float y\nproperty float z\n" + ("property float nx\nproperty float ny\nproperty float nz\n" if has_n else "")
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\ncomment bodyslam_amd TSDF point cloud\nelement vertex {n}\n" + props +
                 "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n").encode("ascii"))
        f.write(rec.tobytes())


def write_ply_mesh(path: str, mesh: TriangleMesh) -> None:
    """binary little-endian PLY of a triangle mesh: vertices x y z (float32) red green blue (uchar), faces as uchar-counted int32 lists
    -- what o3d.io.write_triangle_mesh writes for a .ply path (Open3D stores coordinates as doubles; the values here are fp32)"""
    nv, nt = mesh.vertices.shape[0], mesh.triangles.shape[0]
    vrec = np.empty(nv, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    vrec["x"], vrec["y"], vrec["z"] = mesh.vertices[:, 0], mesh.vertices[:, 1], mesh.vertices[:, 2]
    c = np.clip(np.rint(mesh.vertex_colors * 255.0), 0, 255).astype(np.uint8) if nv else np.zeros((0, 3), np.uint8)
    vrec["red"], vrec["green"], vrec["blue"] = c[:, 0], c[:, 1], c[:, 2]
    frec = np.empty(nt, dtype=[("n", "u1"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])
    frec["n"] = 3
    if nt:
        frec["a"], frec["b"], frec["c"] = mesh.triangles[:, 0], mesh.triangles[:, 1], mesh.triangles[:, 2]
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\ncomment bodyslam_amd TSDF triangle mesh\nelement vertex {nv}\n"
                 "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                 f"element face {nt}\nproperty list uchar int vertex_indices\nend_header\n").encode("ascii"))
        f.write(vrec.tobytes())
        f.write(frec.tobytes())
