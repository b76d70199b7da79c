"""Multi-process (gloo, world_size 2, CPU) coverage of the N>1 path: block bounds, pair ownership with the
one-frame halo, and the relative-pose all-gather that stitches the chain."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bodyslam_amd.pipeline import gather_relative_poses, local_pairs, shard_bounds
from oracle import geom3d_ref as G


def test_shard_bounds_cover_and_balance():
    for n in (1, 2, 7, 256, 1000, 1001, 4000):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in blocks]
            assert max(sizes) - min(sizes) <= 1
            pairs = np.concatenate([local_pairs(s, e) for s, e in blocks]) if n > 1 else np.zeros((0, 2), np.int32)
            # every consecutive pair exactly once, in order: N frames -> N-1 relatives (MPEM_eval.py:216-223)
            assert np.array_equal(pairs, np.stack([np.arange(n - 1), np.arange(1, n)], 1).astype(np.int32))


def test_single_frame_and_empty_blocks():
    assert local_pairs(0, 1).shape == (0, 2)
    assert shard_bounds(1, 2, 1) == (1, 1)
    assert local_pairs(1, 1).shape == (0, 2)


def _worker(rank, world, port, n_frames, golden, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t_rel = torch.from_numpy(np.load(golden)["t_rel"][: n_frames - 1]).reshape(-1, 16)
    start, end = shard_bounds(n_frames, world, rank)
    pg = local_pairs(start, end)
    mine = t_rel[pg[:, 1] - 1] if pg.shape[0] else torch.zeros(0, 16)
    counts = [local_pairs(*shard_bounds(n_frames, world, r)).shape[0] for r in range(world)]
    full = gather_relative_poses(mine.contiguous(), counts)
    ok = torch.equal(full, t_rel)
    g = G.pose_chain(full.numpy().reshape(-1, 4, 4))
    np.save(os.path.join(out_dir, f"g_{rank}.npy"), g)
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [11, 64, 257])
def test_all_gather_stitches_the_chain(tmp_path, golden_dir, n_frames):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    golden = os.path.join(golden_dir, "geom3d_chain.npz")
    mp.spawn(_worker, args=(2, port, n_frames, golden, str(tmp_path)), nprocs=2, join=True)
    ref = np.load(golden)["g_abs"][:n_frames]
    for r in range(2):
        assert open(tmp_path / f"ok_{r}").read() == "1"
        # every rank evaluates the identical chain: bit-equal to the single-process reference result
        assert np.array_equal(np.load(tmp_path / f"g_{r}.npy"), ref)
