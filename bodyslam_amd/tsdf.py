"""TSDF map of the SLAM loop: drop-in for BodySLAM_not_refactored/3DM/tsdf.py:5-52.

The reference's ``TSDF`` wraps Open3D's ``ScalableTSDFVolume(voxel_length=0.001, sdf_trunc=0.1, RGB8, volume_unit_resolution=32,
depth_sampling_stride=8)``: ``build_3D_map(rgbd, intrinsic, extrinsic)`` integrates a frame (3DM/slam.py:117,179 -- it passes the
accumulated pose as ``extrinsic``), ``extract_pcd`` / ``save_pcd`` read the surface points back (:126,191,195).  Same class, method
names and defaults here.  The voxels live in HBM (one contiguous block of res^3 x 5 fp32 per volume unit, allocated in slabs and
zero-filled once) and are integrated / extracted by the HIP kernels of csrc/tsdf.hip; which units exist is host state (a dict, as
Open3D's unordered_map), found per frame from the strided depth sample exactly as Open3D does.  With the reference's parameters a
unit is 3.2 cm wide and a point opens the ~7^3 units within 0.1 m of it; a 640x480 frame of a surface at 12 cm touches ~1 600 units
= 1 GB of voxel state (measured, tools/probes/tsdf_full_size.py: kernel 0.33 ms, the numpy unit discovery ~90 ms -- the step is
host-bound), metre-scale scenes proportionally more: the voxel store is sized for 288 GB of HBM.  Open3D is not vendored and not installable offline: parity against it is unpinned (oracle/tsdf_ref.py
restates the same algorithm in numpy; tests/ compare the two).  Not built: normals of the extracted points and
``extract_mesh`` / ``save_mesh`` (marching cubes) -- they raise NotImplementedError.
"""
from __future__ import annotations

import copy
import ctypes as C
from dataclasses import dataclass
from typing import Optional

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
    points: np.ndarray      # [M, 3] float32
    colors: np.ndarray      # [M, 3] float32 in [0, 1]


_OFF = 1 << 20              # unit indices are packed as three 21-bit fields


def _pack(ix, iy, iz):
    return ((ix + _OFF).astype(np.int64) << 42) | ((iy + _OFF).astype(np.int64) << 21) | (iz + _OFF).astype(np.int64)


class TSDF:
    def __init__(self, voxel_length: float = 0.001, sdf_trunc: float = 0.1, volume_unit_resolution: int = 32,
                 depth_sampling_stride: int = 8, device: int = 0, slab_bytes: int = 1 << 30):
        self.voxel_length, self.sdf_trunc = float(voxel_length), float(sdf_trunc)
        self.res, self.stride = int(volume_unit_resolution), int(depth_sampling_stride)
        self.unit_length = self.voxel_length * self.res
        self.dev = torch.device("cuda", device)
        L.init(device)
        self.unit_floats = self.res ** 3 * 5
        self.slab_units = max(1, slab_bytes // (self.unit_floats * 4))
        self.slabs = []                 # fp32 [slab_units, unit_floats] tensors, zero-filled
        self.slot = {}                  # packed unit index -> slot
        self.index = []                 # slot -> (ix, iy, iz)

    # ---- the reference's surface -----------------------------------------------------------------
    def build_3D_map(self, rgbd: RGBDImage, intrinsic: PinholeCameraIntrinsic, extrinsic) -> None:
        depth = np.ascontiguousarray(np.asarray(rgbd.depth, dtype=np.float32))
        H, W = depth.shape
        E = np.ascontiguousarray(np.asarray(self._np(extrinsic), dtype=np.float64))
        K = np.array([intrinsic.fx, intrinsic.fy, intrinsic.cx, intrinsic.cy], dtype=np.float64)
        import time
        t0 = time.perf_counter()
        slots = self._touch(depth, K, E)
        self.last_discovery_s = time.perf_counter() - t0            # host side of the step (diagnostics, tools/probes/tsdf_full_size.py)
        self.last_units = int(slots.size)
        if slots.size == 0:
            return
        d_dev = torch.from_numpy(depth).to(self.dev)
        c_dev = None if rgbd.color is None else torch.from_numpy(np.ascontiguousarray(rgbd.color)).to(self.dev)
        idx = torch.from_numpy(np.array([self.index[s] for s in slots], dtype=np.int32)).to(self.dev)
        ptr = torch.from_numpy(self._ptrs(slots)).to(self.dev)
        e12 = np.ascontiguousarray(E[:3].reshape(12))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        L.check(L.load_library().bs_tsdf_integrate(L.p(d_dev), L.p(c_dev), H, W, K.ctypes.data_as(C.c_void_p), e12.ctypes.data_as(C.c_void_p),
                                                   L.p(idx), L.p(ptr), int(slots.size), self.res, self.voxel_length, self.sdf_trunc,
                                                   L.stream_ptr()), "bs_tsdf_integrate")
        ev1.record()
        torch.cuda.current_stream(self.dev).synchronize()          # (the host arrays above must outlive the launch)
        self.last_kernel_ms = ev0.elapsed_time(ev1)

    def build_copy_3D_map(self, rgbd, intrinsic, extrinsic) -> "TSDF":
        other = copy.copy(self)
        other.slabs = [s.clone() for s in self.slabs]
        other.slot, other.index = dict(self.slot), list(self.index)
        other.build_3D_map(rgbd, intrinsic, extrinsic)
        return other

    def extract_pcd(self) -> PointCloud:
        U = len(self.index)
        if U == 0:
            return PointCloud(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
        slots = np.arange(U)
        idx_np = np.array(self.index, dtype=np.int32)
        ptr_np = self._ptrs(slots)
        nbr = np.zeros((U, 3), dtype=np.int64)
        for a in range(3):
            nk = idx_np.copy()
            nk[:, a] += 1
            keys = _pack(nk[:, 0], nk[:, 1], nk[:, 2])
            ns = np.array([self.slot.get(int(k), -1) for k in keys])
            nbr[:, a] = np.where(ns >= 0, ptr_np[np.maximum(ns, 0)], 0)
        idx, ptr, nb = (torch.from_numpy(a).to(self.dev) for a in (idx_np, ptr_np, nbr))
        count = torch.zeros(U, dtype=torch.int32, device=self.dev)
        lib = L.load_library()
        L.check(lib.bs_tsdf_extract(L.p(idx), L.p(ptr), L.p(nb), U, self.res, self.voxel_length, L.p(count), None, None, None, L.stream_ptr()),
                "bs_tsdf_extract")
        counts = count.cpu().numpy().astype(np.int64)
        total = int(counts.sum())
        pts = torch.empty(max(total, 1), 3, device=self.dev)
        cols = torch.empty(max(total, 1), 3, device=self.dev)
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)).to(self.dev)
        L.check(lib.bs_tsdf_extract(L.p(idx), L.p(ptr), L.p(nb), U, self.res, self.voxel_length, L.p(count), L.p(off), L.p(pts), L.p(cols),
                                    L.stream_ptr()), "bs_tsdf_extract")
        return PointCloud(pts[:total].cpu().numpy(), cols[:total].cpu().numpy())

    def save_pcd(self, saving_path: str) -> None:
        write_ply(saving_path, self.extract_pcd())

    def extract_mesh(self):
        raise NotImplementedError("extract_triangle_mesh (marching cubes) is not built; extract_pcd / save_pcd are")

    def save_mesh(self, saving_path: str) -> None:
        self.extract_mesh()

    # ---- host bookkeeping --------------------------------------------------------------------------
    @staticmethod
    def _np(m):
        return m.detach().cpu().numpy() if isinstance(m, torch.Tensor) else np.asarray(m)

    def _ptrs(self, slots: np.ndarray) -> np.ndarray:
        base = np.array([s.data_ptr() for s in self.slabs], dtype=np.int64)
        slots = np.asarray(slots, dtype=np.int64)
        return base[slots // self.slab_units] + (slots % self.slab_units) * (self.unit_floats * 4)

    def _touch(self, depth: np.ndarray, K: np.ndarray, E: np.ndarray) -> np.ndarray:
        """slots of the volume units this frame integrates into (ScalableTSDFVolume::Integrate: every unit that meets the
        +-sdf_trunc box of a point of the strided depth sample; missing ones are opened)"""
        fx, fy, cx, cy = K
        H, W = depth.shape
        ii, jj = np.meshgrid(np.arange(0, H, self.stride), np.arange(0, W, self.stride), indexing="ij")
        z = depth[ii, jj].astype(np.float64)
        m = z > 0
        if not m.any():
            return np.zeros(0, dtype=np.int64)
        z, ii, jj = z[m], ii[m], jj[m]
        cam = np.stack([(jj - cx) * z / fx, (ii - cy) * z / fy, z, np.ones_like(z)], 1)
        p = (cam @ np.linalg.inv(E).T)[:, :3]
        lo = np.floor((p - self.sdf_trunc) / self.unit_length).astype(np.int64)
        hi = np.floor((p + self.sdf_trunc) / self.unit_length).astype(np.int64)
        span = int((hi - lo).max()) + 1
        o = np.stack(np.meshgrid(*(np.arange(span),) * 3, indexing="ij"), -1).reshape(-1, 3)
        keys = []
        step = max(1, (1 << 22) // o.shape[0])          # bound the temporary: ~4 M candidate units at a time
        for s in range(0, p.shape[0], step):
            c = lo[s:s + step, None, :] + o[None]
            ok = (c <= hi[s:s + step, None, :]).all(-1)
            c = c[ok]
            keys.append(np.unique(_pack(c[:, 0], c[:, 1], c[:, 2])))
        keys = np.unique(np.concatenate(keys))
        slots = np.empty(keys.shape[0], dtype=np.int64)
        for n, k in enumerate(keys):
            k = int(k)
            s = self.slot.get(k)
            if s is None:
                s = len(self.index)
                if s >= len(self.slabs) * self.slab_units:
                    self.slabs.append(torch.zeros(self.slab_units, self.unit_floats, device=self.dev))
                self.slot[k] = s
                self.index.append(((k >> 42) - _OFF, ((k >> 21) & ((1 << 21) - 1)) - _OFF, (k & ((1 << 21) - 1)) - _OFF))
            slots[n] = s
        return slots

    def unit(self, key) -> np.ndarray:
        """voxels of one unit as fp32 [res, res, res, 5] (tests, diagnostics)"""
        s = self.slot[int(_pack(*(np.array([v]) for v in key))[0])]
        return self.slabs[s // self.slab_units][s % self.slab_units].view(self.res, self.res, self.res, 5).cpu().numpy()


def write_ply(path: str, pcd: PointCloud) -> None:
    """binary little-endian PLY: x y z (float32), red green blue (uchar) -- the container o3d.io.write_point_cloud produces for a
    .ply path (Open3D stores the coordinates as doubles; the points here are fp32)"""
    n = pcd.points.shape[0]
    rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = pcd.points[:, 0], pcd.points[:, 1], pcd.points[:, 2]
    c = np.clip(np.rint(pcd.colors * 255.0), 0, 255).astype(np.uint8) if n else np.zeros((0, 3), np.uint8)
    rec["red"], rec["green"], rec["blue"] = c[:, 0], c[:, 1], c[:, 2]
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\ncomment bodyslam_amd TSDF point cloud\nelement vertex {n}\n"
                 "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                 "end_header\n").encode("ascii"))
        f.write(rec.tobytes())
