"""The per-frame depth + pose + back-projection loop over a frame sequence, one process per GPU.

What it reproduces (per-stage; the reference runs MDEM offline and the rest in SLAM._sequential_loop):
  depth per frame                 compute_dp / DepthEstimator.infer_depth_map      MDEM/compute_dp.py:8-18
  relative pose per pair (i-1,i)  MPEMInterface.infer_relative_pose_between        EVALUATION/MPEM_eval.py:216-223
  absolute poses                  compute_curr_estimate_global_pose chain          3DM/slam.py:148-153
  points per frame                pixel_to_3d + RGBD constants                     3DM/scaling_system.py:72-77
N frames -> N depth maps, N-1 relatives, N absolute poses (the first is the identity), N point sets.

Multi-GPU: the sequence is cut into contiguous blocks, one per rank.  Rank r
runs MDEM on its frames and MPEM on the pairs (i-1, i) for i in its block (it reads frame start-1 as a
one-frame halo).  ONE RCCL all-gather (torch.distributed backend "nccl" on ROCm) of the per-rank
relative poses [N_r, 16] fp32 stitches the chain; every rank then evaluates the identical fp64 chain
and back-projects its own frames with their absolute poses.  No other collective is on the path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import geom3d
from .cyclepose import CyclePoseEngine
from .zoedepth import ZoeConfig, ZoeDepthEngine


def shard_bounds(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [start, end) of rank `rank`; blocks differ by at most one frame."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def local_pairs(start: int, end: int) -> np.ndarray:
    """Global (prev, curr) index pairs owned by a block: (i-1, i) for i in [max(start,1), end)."""
    i = np.arange(max(start, 1), end, dtype=np.int32)
    return np.stack([i - 1, i], axis=1) if i.size else np.zeros((0, 2), np.int32)


def gather_relative_poses(t_local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """All-gather the per-rank relative poses [P_r,16] into the full [sum P_r,16] in rank order.
    Ranks own different pair counts (rank 0 has one fewer), so blocks are padded to the maximum."""
    import torch.distributed as dist
    if not dist.is_initialized():
        if len(counts) > 1:
            raise RuntimeError(f"gather_relative_poses: {len(counts)} ranks but torch.distributed is not initialised "
                               "(a sharded sequence cannot chain only its local poses)")
        return t_local
    world = dist.get_world_size(group)
    if len(counts) != world or t_local.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError(f"gather_relative_poses: counts {list(counts)} do not match world {world} / local pairs {t_local.shape[0]}")
    pmax = max(counts)
    # RCCL gathers device buffers in place; a gloo group (CPU tests, two processes sharing one GPU) gets the few KB through the host
    cdev = t_local.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = torch.zeros(pmax, 16, dtype=torch.float32, device=cdev)
    buf[: t_local.shape[0]] = t_local
    out = torch.empty(world * pmax, 16, dtype=torch.float32, device=cdev)
    if cdev.type == "cpu":
        dist.all_gather(list(out.view(world, pmax, 16).unbind(0)), buf, group=group)
    else:
        dist.all_gather_into_tensor(out, buf, group=group)
    return torch.cat([out[r * pmax: r * pmax + counts[r]] for r in range(world)], dim=0).to(t_local.device)


def share_calibration(make_report: Callable[[], dict], group=None, device=None) -> dict:
    """Every rank of a sharded run must evaluate the same arithmetic: rank 0 calibrates (``make_report()``, e.g.
    ``ZoeDepthEngine.calibrate``) and broadcasts its report; the other ranks return it without calibrating themselves (ranks
    calibrating on their own could fall on different sides of a tolerance).  Not on the data path: once per engine, a few hundred
    bytes.  Without an initialised process group (one GPU) the report is simply made."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return make_report()
    rank = dist.get_rank(group)
    # rank 0's failure (out of memory while the reference engine is up, a launch error) must not leave the others waiting in the
    # broadcast: the outcome travels as (ok, report-or-message) and every rank raises (round-4 advisor)
    box, err = [None], None
    if rank == 0:
        try:
            box = [(True, make_report())]
        except Exception as e:      # noqa: BLE001 -- re-raised below, on every rank
            err = e
            box = [(False, f"{type(e).__name__}: {e}")]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group, device=device)
    ok, payload = box[0]
    if not ok:
        if err is not None:
            raise err
        raise RuntimeError(f"share_calibration: rank 0 failed to calibrate: {payload}")
    return payload


@dataclass
class SequenceResult:
    start: int
    end: int
    depth_u16: torch.Tensor            # int16 storage of uint16 metres*256, [n_local, H, W]
    t_rel: torch.Tensor                # fp32 [N-1, 4, 4]  (full sequence, identical on every rank)
    g_abs: torch.Tensor                # fp64 [N, 4, 4]    (full sequence, identical on every rank)
    point_counts: torch.Tensor         # int32 [n_local]
    points: Optional[list] = None      # optional [(xyz [M,3] fp32, idx [M] int32)] per local frame
    depth_m: Optional[torch.Tensor] = None
    tsdf: Optional[object] = None      # run_slam_loop: the map as the loop left it -- NOT necessarily the volume the caller passed in: a rebuild
                                       # (pose graph moved the poses, or frame 2000) starts a fresh one, as slam.py:159-185 does


class BodySlamPipeline:
    def __init__(self, zoe_weights: Dict[str, torch.Tensor], pose_weights: Dict[str, torch.Tensor],
                 zoe_cfg: Optional[ZoeConfig] = None, dtype=torch.float16, device: int = 0, batch: int = 8,
                 K: Sequence[float] = geom3d.REF_INTRINSICS, depth_scale: float = geom3d.REF_DEPTH_SCALE,
                 depth_trunc: float = geom3d.REF_DEPTH_TRUNC, flip_aug: bool = True,
                 target_hw: Tuple[int, int] = (384, 512), precision: str = "accurate", pad_ragged: bool = True):
        """precision: ZoeDepthEngine's -- "accurate" keeps depth within 1e-4 m (L1) of the fp32 reference, "fast" is one
        16-bit MFMA pass per product (L1 ~3e-4 m at fp16), "reference" three 16-bit passes on (hi | lo) pairs for every product
        (~1e-5 m; the precision calibrate() measures the others against, and the one bf16 storage needs to meet the tolerance).  pad_ragged: run a ragged last batch of a block through the
        full-size plan instead of building a second plan for its size."""
        L.init(device)
        self.dev = torch.device("cuda", device)
        self.batch = batch
        self.pad_ragged = pad_ragged
        # BASELINE config 5 ("with pose-graph re-linearisation"): the reference optimises its pose graph every 500 frames
        # (3DM/slam.py:54,159-165).  0 = off.  loop_closures: extra (source, target, T[4,4], information[6,6]) edges, uncertain
        # (the reference never adds any: slam.py:30,80)
        self.posegraph_every = 0
        self.loop_closures = []
        self.K, self.depth_scale, self.depth_trunc, self.flip = tuple(K), depth_scale, depth_trunc, flip_aug
        self.zoe = ZoeDepthEngine(zoe_weights, zoe_cfg, dtype=dtype, device=device, target_hw=target_hw, precision=precision)
        self.precision = precision
        self.pose = CyclePoseEngine(pose_weights, dtype=dtype, device=device, precision="accurate" if precision == "reference" else precision)

    def calibrate(self, H: int, W: int, group=None) -> Optional[dict]:
        """the depth engine's load-time calibration (ZoeDepthEngine.calibrate), made once by rank 0 and shared (share_calibration)"""
        z = self.zoe
        if not (z.acc and z.auto_modes):
            return z.calibration
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            if z.calibration is None:
                z.calibrate(H, W)
            return z.calibration
        # whether the collective below is entered is decided from what ALL ranks hold, never from this rank's state alone (a rank that
        # was calibrated earlier, or had a report applied, would otherwise skip a broadcast the others wait in: round-4 advisor)
        cdev = self.dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
        need = torch.tensor([1 if z.calibration is None else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(need, op=dist.ReduceOp.MAX, group=group)
        if int(need.item()):
            first = dist.get_rank(group) == 0
            rep = share_calibration(lambda: z.calibration if z.calibration is not None else z.calibrate(H, W), group, cdev)
            if not first:
                z.apply_calibration(rep)        # (also on a rank that held one: every rank runs rank 0's arithmetic)
        return z.calibration

    # -- stage 1+2 for one block of frames ----------------------------------------------------------
    def depth_and_pose_block(self, frames: torch.Tensor, start: int, end: int, keep_depth_m: bool = False, frame_offset: int = 0,
                             pad_to_batch: bool = False):
        """frames: the sequence as a uint8 tensor [N,H,W,3] (CPU pinned or GPU); processes the frames [start,end) (global
        indices).  frame_offset: global index of frames[0] -- a rank may hold only its block and the one-frame halo."""
        _, H, W, _ = frames.shape
        assert frame_offset <= max(start - 1, 0) and end - frame_offset <= frames.shape[0], (frame_offset, start, end, frames.shape[0])
        n_local = end - start
        depth = torch.empty(n_local, H, W, dtype=torch.int16, device=self.dev)
        depth_m = torch.empty(n_local, H, W, dtype=torch.float32, device=self.dev) if keep_depth_m else None
        pairs_g = local_pairs(start, end)
        t_rel = torch.empty(pairs_g.shape[0], 16, dtype=torch.float32, device=self.dev)
        pi = 0
        for b0 in range(start, end, self.batch):
            b1 = min(b0 + self.batch, end)
            h0 = max(b0 - 1, 0)                                     # one-frame halo for the first pair of the batch
            chunk = frames[h0 - frame_offset: b1 - frame_offset].to(self.dev, non_blocking=True)
            nb = b1 - b0
            if self.pad_ragged and nb < self.batch and (end - start > self.batch or pad_to_batch):
                # a ragged last batch runs through the full-size plan (its buffers exist already; a second plan of nearly the
                # same size would cost its own tens of GB): the missing frames repeat the last one and their outputs are
                # dropped.  Results do not depend on the batch (per-image routing; tests/test_fullsize_properties_gpu.py).
                fr = chunk[b0 - h0:]
                fr = torch.cat([fr, fr[-1:].expand(self.batch - nb, -1, -1, -1)], 0)
                dm, du = self.zoe.infer(fr, flip_aug=self.flip)
                dm, du = dm[:nb], du[:nb]
            else:
                dm, du = self.zoe.infer(chunk[b0 - h0:], flip_aug=self.flip)
            depth[b0 - start: b1 - start].copy_(du)
            if keep_depth_m:
                depth_m[b0 - start: b1 - start].copy_(dm)
            i = np.arange(max(b0, 1), b1, dtype=np.int32)
            if i.size:
                pl = torch.from_numpy(np.stack([i - 1 - h0, i - h0], axis=1).astype(np.int32)).to(self.dev)
                T = self.pose.infer_pairs(chunk, pl)
                t_rel[pi: pi + i.size].copy_(T.reshape(-1, 16))
                pi += i.size
        return depth, depth_m, t_rel

    def _posegraph_relinearise(self, g_abs: torch.Tensor, t_all: torch.Tensor) -> torch.Tensor:
        """3DM/slam.py:156-175 on the finished chain: nodes = absolute poses, one odometry edge per frame, global optimisation
        every ``posegraph_every`` nodes; replicated on every rank (host side, as in the reference).  With odometry edges only
        the chain is the optimum and comes back bit for bit."""
        from .posegraph import PoseGraph, update_global_extrinsic
        G = g_abs.cpu().numpy()
        T = t_all.view(-1, 4, 4).cpu().numpy().astype(np.float64)
        N = G.shape[0]
        pg = PoseGraph()
        pg.add_node(G[0])
        lc = sorted(self.loop_closures, key=lambda e: max(e[0], e[1]))
        k = 0
        changed = False
        for i in range(1, N):
            # (after an optimisation moved the nodes, the chain continues from the updated last pose, slam.py:148-153,165)
            pg.add_node(G[i] if not changed else geom3d.compute_curr_estimate_global_pose(update_global_extrinsic(pg.pose_graph)[-1],
                                                                                          T[i - 1].astype(np.float32)))
            pg.add_edge(T[i - 1], i, i - 1, False)
            while k < len(lc) and max(lc[k][0], lc[k][1]) <= i:
                pg.add_edge(lc[k][2], lc[k][0], lc[k][1], True, lc[k][3])
                k += 1
            if i % self.posegraph_every == 0:                    # (the reference optimises at i % num_posegraph_optim == 0 only, slam.py:159)
                before = [n.pose.copy() for n in pg.pose_graph.nodes]
                pg.optimize()
                changed = changed or any(not np.array_equal(a, n.pose) for a, n in zip(before, pg.pose_graph.nodes))
        if not changed:
            return g_abs
        return torch.from_numpy(np.stack(update_global_extrinsic(pg.pose_graph))).to(g_abs.device)

    def run_sequence(self, frames, rank: int = 0, world: int = 1, group=None, keep_points: bool = False,
                     keep_depth_m: bool = False, on_points: Optional[Callable] = None,
                     gather: Optional[Callable] = None, frame_offset: int = 0, n_frames: Optional[int] = None) -> SequenceResult:
        """``gather(t_local, counts) -> t_all`` replaces the RCCL all-gather (tests emulate several ranks on one GPU with
        it); by default the relatives go through ``gather_relative_poses`` on ``group``.  A rank that holds only its block
        (+ the halo frame) passes it with ``frame_offset`` = global index of frames[0] and ``n_frames`` = sequence length."""
        frames = torch.as_tensor(frames)
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3
        N = frames.shape[0] if n_frames is None else n_frames
        start, end = shard_bounds(N, world, rank)
        if world > 1 and gather is None:
            self.calibrate(int(frames.shape[1]), int(frames.shape[2]), group)
        depth, depth_m, t_local = self.depth_and_pose_block(frames, start, end, keep_depth_m, frame_offset)
        counts = [local_pairs(*shard_bounds(N, world, r)).shape[0] for r in range(world)]
        if gather is not None:
            t_all = gather(t_local, counts)
        else:
            t_all = gather_relative_poses(t_local, counts, group) if world > 1 else t_local
        return self.chain_and_backproject(N, start, end, depth, depth_m, t_all, keep_points, on_points)

    def run_slam_loop(self, frames, vo: bool = False, tsdf=None, keep_points: bool = False, keep_depth_m: bool = False,
                      posegraph_every: Optional[int] = None, rebuild_every: int = 2000, extract_every_frame: bool = False,
                      tsdf_factory: Optional[Callable] = None, on_frame: Optional[Callable] = None) -> SequenceResult:
        """The reference's whole per-frame loop (``SLAM._first_loop`` / ``_sequential_loop``, 3DM/slam.py:95-205) on one GPU, streamed:
        the network stages run a batch of frames ahead (they do not depend on the loop's state), everything that does runs frame by
        frame in the reference's order --

          i = 0     identity pose, pose-graph node, map integration                                              slam.py:95-127
          i >= 1    MPEM relative pose; with ``vo``: RGB-D odometry between the two pseudo-RGBD frames (raw depth / depth_scale, NOT
                    truncated: slam_utils.py:187,228 builds rgbd_t from the raw images) -> 3-state UKF -> the translation replaced by
                    the filter state (visual_odometry.py:60-93); fp64 chain step (slam_utils.py:110-122); pose-graph node + odometry
                    edge (slam.py:156-157)
                    if i % posegraph_every == 0 (the reference: 500): optimise, read the node poses back, and -- only when they
                      moved -- rebuild the whole map from frames 0..i with the new poses (update_map_after_pg); frame i is NOT
                      integrated on its own in this branch (slam.py:159-175)
                    else: integrate frame i with its pose (slam.py:179)
                    if i % rebuild_every == 0 (2000): rebuild the map (slam.py:183-185)
                    extract_pcd (every frame in the reference, slam.py:195; here behind ``extract_every_frame``)

        and nothing of it waits for the GPU per frame: the odometry of a batch is tracked on the device and read back once
        (``RGBDOdometry.track``), the UKF / chain / pose graph are host arithmetic on a few numbers (as in the reference), the map
        steps are enqueued without a round trip (``TSDF.build_3D_map(sync=False)``; block capacity is reserved a batch ahead and
        checked at the batch end).  ``tsdf_factory()`` makes the fresh TSDF of a rebuild (default: a copy of ``tsdf``'s parameters).
        Returns the poses as they stand at the end (after pose-graph updates); ``on_frame(i, pose, pcd_or_None)`` is called per frame."""
        from .posegraph import PoseGraph, update_global_extrinsic
        from .tsdf import PinholeCameraIntrinsic, RGBDImage, TSDF
        frames = torch.as_tensor(frames)
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3
        N, H, W, _ = frames.shape
        every = self.posegraph_every if posegraph_every is None else posegraph_every
        fr_dev = frames.to(self.dev)
        depth_all = torch.empty(N, H, W, dtype=torch.int16, device=self.dev)
        depth_m_all = torch.empty(N, H, W, dtype=torch.float32, device=self.dev) if keep_depth_m else None
        intr = PinholeCameraIntrinsic(W, H, *[float(v) for v in self.K])
        odo = None
        if vo:
            from .rgbd_odometry import RGBDOdometry
            from .visual_odometry import VO
            odo = RGBDOdometry(tuple(float(v) for v in self.K), device=self.dev.index or 0)
            stored = {}

            class _Batched:                          # MPEM already ran batched: hand VO its result for the pair it asks about
                def infer_relative_pose_between(self_, prev, curr):
                    return stored["mpem"]
            vo_obj = VO(_Batched(), intrinsic=tuple(float(v) for v in self.K), rgbd_odometry=lambda c, p: stored["odo"])
        if tsdf is not None and tsdf_factory is None:
            tsdf_factory = lambda: TSDF(tsdf.voxel_length, tsdf.sdf_trunc, tsdf.res, tsdf.stride, device=self.dev.index or 0,
                                        slab_bytes=tsdf.slab_units * tsdf.unit_floats * 4, max_units=tsdf.max_units)
        state = {"tsdf": tsdf}

        def tsdf_depth(j0, j1):
            """pseudo-RGBD depth of the map step (slam_utils.py:212-220): depth / depth_scale, values >= depth_trunc dropped"""
            return L.depth_u16_to_m(depth_all[j0:j1].contiguous(), self.depth_scale, self.depth_trunc)

        def rebuild(upto, poses):
            """update_map_after_pg (slam_utils.py:124-135): a fresh volume, frames 0..upto integrated with the current poses"""
            t = tsdf_factory()
            if state["tsdf"] is not None:
                t.reserve(state["tsdf"].n_units_known())
            for j0 in range(0, upto + 1, self.batch):
                j1 = min(j0 + self.batch, upto + 1)
                dm = tsdf_depth(j0, j1)
                t.build_3D_map_batch([RGBDImage(fr_dev[j], dm[j - j0]) for j in range(j0, j1)], intr, poses[j0:j1])
            t.sync()
            state["tsdf"] = t

        def run_map_actions(actions, b0, dm_map):
            """the map steps a batch of frames asked for, in order.  A run of plain integrations is ONE pass over the map
            (``TSDF.build_3D_map_batch``: unit discovery of all its frames, one round trip that makes exactly the blocks they need, every
            touched voxel loaded once and updated with its frames in order) -- bit for bit the frame-by-frame result."""
            k = 0
            while k < len(actions):
                if actions[k][0] == "rebuild":
                    rebuild(actions[k][1], actions[k][2])
                    k += 1
                    continue
                run = []
                while k < len(actions) and actions[k][0] == "int":
                    run.append(actions[k])
                    k += 1
                state["tsdf"].build_3D_map_batch([RGBDImage(fr_dev[i], dm_map[i - b0]) for (_, i, _) in run], intr, [pose for (_, _, pose) in run])

        pg = PoseGraph()
        extr, rel_fused = [], []
        cnt_all = torch.empty(N, dtype=torch.int32, device=self.dev)
        points = [] if keep_points else None
        # Two streams.  The network stages of batch k + 1 (MDEM + MPEM: ~180 ms of large GEMMs at B = 64) are enqueued on the caller's
        # stream BEFORE batch k's sequential part runs -- odometry tracking, UKF, chain, pose graph, map steps: small latency-bound
        # kernels and host arithmetic -- which goes to a side stream that waits only for batch k's network results.  The host work and
        # the small kernels hide under the next batch's GEMMs (the round-2 loop ran them one after the other: 72 frames/s).
        main = torch.cuda.current_stream(self.dev)
        if getattr(self, "_loop_stream", None) is None:
            self._loop_stream = torch.cuda.Stream(device=self.dev, priority=-1)   # its short kernels go ahead of the network's big ones
        side = self._loop_stream

        def launch_network(b0):
            b1 = min(b0 + self.batch, N)
            d_u16, d_m, t_dev = self.depth_and_pose_block(fr_dev, b0, b1, keep_depth_m, 0, pad_to_batch=N > self.batch)      # (the device copy: no re-upload per batch)
            depth_all[b0:b1].copy_(d_u16)
            if keep_depth_m:
                depth_m_all[b0:b1].copy_(d_m)
            t_dev.record_stream(side)
            ev = torch.cuda.Event()
            ev.record(main)
            return b0, b1, t_dev, ev

        pending = launch_network(0)
        while pending is not None:
            b0, b1, t_dev, ev = pending
            pending = launch_network(b1) if b1 < N else None                    # batch k + 1's network: enqueued behind batch k's, not waited for
            side.wait_event(ev)
            with torch.cuda.stream(side):
                t_mpem = t_dev.view(-1, 4, 4).cpu().numpy()                     # the pairs (i - 1, i), i in [max(b0, 1), b1); waits for batch k's network only
                self._slam_batch(b0, b1, t_mpem, vo, odo, stored if vo else None, vo_obj if vo else None, state, fr_dev, depth_all, intr, pg, extr,
                                 rel_fused, cnt_all, points, keep_points, every, rebuild_every, extract_every_frame, on_frame, tsdf_depth,
                                 run_map_actions)
        main.wait_stream(side)
        self.last_tsdf = state["tsdf"]
        t_rel = torch.from_numpy(np.stack(rel_fused).astype(np.float32)).to(self.dev) if rel_fused else torch.zeros(0, 4, 4, device=self.dev)
        g_abs = torch.from_numpy(np.stack(self._extr_final)).to(self.dev)
        # (points / point_counts were back-projected batch by batch with the poses as they stood then; g_abs holds the poses after the last
        # pose-graph step -- the reference's per-frame point clouds have the same property, slam.py:195.  The fused relatives are fp32 as in
        # the reference: visual_odometry.py:90 writes the filter state into MPEM's float32 matrix.)
        return SequenceResult(0, N, depth_all, t_rel, g_abs, cnt_all, points, depth_m_all, state["tsdf"])

    def _slam_batch(self, b0, b1, t_mpem, vo, odo, stored, vo_obj, state, fr_dev, depth_all, intr, pg, extr, rel_fused, cnt_all, points,
                    keep_points, every, rebuild_every, extract_every_frame, on_frame, tsdf_depth, run_map_actions):
        """the sequential part of run_slam_loop for the frames [b0, b1) (see there); runs on the loop's side stream"""
        from .posegraph import update_global_extrinsic
        from .tsdf import RGBDImage
        if True:
            t_odo = None
            if vo:
                # the odometry's depth: raw / depth_scale (no truncation); the pairs of a batch do not depend on each other
                raw = L.depth_u16_to_m(depth_all[b0:b1].contiguous(), self.depth_scale, 3.0e38)
                got = odo.track_block(fr_dev[b0:b1], raw)                      # the block's pairs at once, on the device
                if got.shape[0]:
                    t12 = got.cpu().numpy().reshape(-1, 3, 4)                   # ONE readback per batch
                    t_odo = np.tile(np.eye(4), (t12.shape[0], 1, 1))
                    t_odo[:, :3] = t12
                    t_odo = np.linalg.inv(t_odo)                                # what _compute_vo_o3d returns (visual_odometry.py:118)
            dm_map = tsdf_depth(b0, b1) if state["tsdf"] is not None else None
            actions = []                              # the batch's map steps; executed together unless a point cloud is wanted per frame
            first_pair = max(b0, 1)
            # the fused relatives of the batch: the UKF is sequential over the frames but does not look at the poses
            fused = {}
            for i in range(first_pair, b1):
                T = np.array(t_mpem[i - first_pair])
                if vo:
                    stored["mpem"], stored["odo"] = T, t_odo[i - first_pair]
                    T = vo_obj.estimate_relative_pose_between(i - 1, i, None, None, i)
                fused[i] = T
            ahead = {}                                # poses chained ahead on the device, from extr[-1] up to the next frame on which the pose graph may move them

            def chained_pose(i):
                """slam_utils.py:110-122 for frame i; one device call and one readback per segment of the batch instead of per frame"""
                if i not in ahead:
                    j1 = i
                    while j1 + 1 < b1 and not (every > 0 and j1 % every == 0):
                        j1 += 1
                    seg = geom3d.pose_chain(np.stack([fused[j] for j in range(i, j1 + 1)]).astype(np.float32), g0=np.asarray(extr[-1], dtype=np.float64),
                                            device=self.dev.index or 0).cpu().numpy()
                    ahead.clear()
                    for j in range(i, j1 + 1):
                        ahead[j] = seg[j - i + 1]
                return ahead.pop(i)

            for i in range(b0, b1):
                pcd = None
                if i == 0:
                    pose = np.identity(4, dtype=np.float64)
                    extr.append(pose)
                    pg.add_node(pose)
                    if state["tsdf"] is not None:
                        actions.append(("int", 0, pose))
                else:
                    T = fused[i]
                    rel_fused.append(T)
                    pose = chained_pose(i)
                    extr.append(pose)
                    pg.add_node(pose)
                    pg.add_edge(T, i, i - 1, False)
                    for (s_, t_, Tl, info) in self.loop_closures:               # (the reference never adds any: slam.py:30,80)
                        if max(s_, t_) == i:
                            pg.add_edge(Tl, s_, t_, True, info)
                    if every > 0 and i % every == 0:
                        before = [p_.copy() for p_ in extr]
                        pg.optimize()
                        extr[:] = update_global_extrinsic(pg.pose_graph)        # (in place: the caller's list)
                        if state["tsdf"] is not None and not all(np.array_equal(a_, b_) for a_, b_ in zip(before, extr)):
                            actions.append(("rebuild", i, [p_.copy() for p_ in extr]))
                    elif state["tsdf"] is not None:
                        actions.append(("int", i, extr[-1].copy()))
                    if state["tsdf"] is not None and rebuild_every > 0 and i % rebuild_every == 0:
                        actions.append(("rebuild", i, [p_.copy() for p_ in extr]))
                if extract_every_frame and state["tsdf"] is not None:
                    run_map_actions(actions, b0, dm_map)                        # the map as it stands after frame i (slam.py:195)
                    actions = []
                    state["tsdf"].sync()
                    pcd = state["tsdf"].extract_pcd()
                if on_frame is not None:
                    on_frame(i, extr[-1], pcd)
            if state["tsdf"] is not None:
                run_map_actions(actions, b0, dm_map)
                state["tsdf"].sync()                                            # unit counts of the batch, overflow check
            # back-projection of the batch with the poses as they stand (the hot path's D3 output; the reference's loop keeps the map only)
            g_b = torch.from_numpy(np.stack(extr[b0:b1])).to(self.dev)
            xyz, idx, cnt = geom3d.backproject(depth_all[b0:b1], self.K, self.depth_scale, self.depth_trunc, poses=g_b)
            cnt_all[b0:b1].copy_(cnt)
            if keep_points:
                c = cnt.cpu().tolist()
                for j in range(b1 - b0):
                    points.append((xyz[j, :c[j]].clone(), idx[j, :c[j]].clone()))
        self._extr_final = list(extr)

    def fuse_vo(self, frames, depth_u16: torch.Tensor, t_rel: torch.Tensor) -> torch.Tensor:
        """visual_odometry.py:60-93 over the pairs (i-1, i) in order: returns t_rel with each translation replaced by the UKF state"""
        from .visual_odometry import VO

        class _Batched:                              # MPEM already ran batched: hand VO its result for the pair it asks about
            def __init__(self, t):
                self.t, self.i = t, 0

            def infer_relative_pose_between(self, prev, curr):
                return self.t[self.i]

        from .tsdf import RGBDImage
        t_np = t_rel.view(-1, 4, 4).cpu().numpy().copy()
        mp = _Batched(t_np)
        vo = VO(mp, intrinsic=tuple(float(v) for v in self.K))
        fr = frames.to(self.dev)                                                   # images and depth stay on the device
        dm = L.depth_u16_to_m(depth_u16.contiguous(), self.depth_scale, self.depth_trunc)
        out = np.empty_like(t_np)
        for i in range(1, fr.shape[0]):
            mp.i = i - 1
            out[i - 1] = vo.estimate_relative_pose_between(i - 1, i, RGBDImage(fr[i - 1], dm[i - 1]), RGBDImage(fr[i], dm[i]), i)
        return torch.from_numpy(out).to(t_rel.device, t_rel.dtype).view(t_rel.shape)

    def integrate_tsdf(self, tsdf, frames, result: SequenceResult) -> None:
        """The plain map step (3DM/slam.py:117,179) for a finished ``run_sequence``: EVERY local frame's pseudo-RGBD
        (3DM/slam_utils.py:212-220: depth / depth_scale, values >= depth_trunc dropped) goes into `tsdf` (bodyslam_amd.tsdf.TSDF) with the
        frame's absolute pose as the extrinsic argument.  The reference's interleaving with the pose graph -- frames on which it
        optimises are not integrated, a moved graph rebuilds the map -- is ``run_slam_loop``'s."""
        from .tsdf import PinholeCameraIntrinsic, RGBDImage
        fr = torch.as_tensor(frames).to(self.dev)                                 # images and depth stay on the device
        dm = L.depth_u16_to_m(result.depth_u16.contiguous(), self.depth_scale, self.depth_trunc)
        H, W = dm.shape[1:]
        intr = PinholeCameraIntrinsic(W, H, *[float(v) for v in self.K])
        g = result.g_abs.cpu().numpy()
        for j in range(dm.shape[0]):
            tsdf.build_3D_map(RGBDImage(fr[result.start + j], dm[j]), intr, g[result.start + j])

    def chain_and_backproject(self, N: int, start: int, end: int, depth: torch.Tensor, depth_m: Optional[torch.Tensor],
                              t_all: torch.Tensor, keep_points: bool = False, on_points: Optional[Callable] = None) -> SequenceResult:
        """Stage 3 of a rank: the replicated fp64 chain over the gathered relatives, then the rank's own back-projection."""
        if t_all.shape[0] != max(N - 1, 0):
            raise ValueError(f"chain needs the {max(N - 1, 0)} relatives of the whole sequence, got {t_all.shape[0]}")
        g_abs = geom3d.pose_chain(t_all, device=self.dev.index or 0)          # [N,4,4] fp64, replicated
        if self.posegraph_every > 0 and N > 1:
            g_abs = self._posegraph_relinearise(g_abs, t_all)
        n_local = end - start
        cnt_all = torch.empty(n_local, dtype=torch.int32, device=self.dev)
        points = [] if keep_points else None
        for b0 in range(0, n_local, self.batch):
            b1 = min(b0 + self.batch, n_local)
            xyz, idx, cnt = geom3d.backproject(depth[b0:b1], self.K, self.depth_scale, self.depth_trunc,
                                               poses=g_abs[start + b0: start + b1])
            cnt_all[b0:b1].copy_(cnt)
            if on_points is not None:
                on_points(start + b0, xyz, idx, cnt)
            if keep_points:
                c = cnt.cpu().tolist()
                for j in range(b1 - b0):
                    points.append((xyz[j, :c[j]].clone(), idx[j, :c[j]].clone()))
        return SequenceResult(start, end, depth, t_all.view(-1, 4, 4), g_abs, cnt_all, points, depth_m)
