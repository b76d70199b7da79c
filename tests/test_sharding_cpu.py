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


def test_make_sequence_at_is_one_sequence_whatever_the_subset():
    """bench.py at N > 1: every rank synthesises only its own frames of ONE sequence -- frame i must not depend on which other frames
    are asked for, and the blocks (B frames + the one-frame halo) of consecutive ranks must overlap in exactly that halo frame"""
    import numpy as np
    from bodyslam_amd.synthetic import make_sequence_at
    total, h, w, B, world = 41, 24, 32, 4, 2
    whole = make_sequence_at(range(total), total, h, w, seed=3)
    some = make_sequence_at([7, 0, 40, 8], total, h, w, seed=3)
    assert np.array_equal(some, whole[[7, 0, 40, 8]])
    assert not np.array_equal(whole[7], whole[8])
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                                            # (bench.py's own index helpers: what a rank of the driver's N > 1 run uses)
    blocks = {}
    for rank in range(world):
        tot, idx = bench.weak_frame_indices(5, B, world, rank)
        assert tot == total and len(idx) == 5 * (B + 1)
        blocks[rank] = (idx, make_sequence_at(idx, total, h, w, seed=3))
    # step k of rank r: its chunk is the halo + B consecutive frames, and the ranks' blocks of a step are consecutive in the sequence
    for k in range(5):
        for rank in range(world):
            ch = blocks[rank][0][bench.weak_chunk(k, B)]
            assert ch == list(range((k * world + rank) * B, (k * world + rank) * B + B + 1))
    i0, f0 = blocks[0]
    i1, f1 = blocks[1]
    assert i1[0] == i0[B] and np.array_equal(f1[0], f0[B])                 # rank 1's halo = rank 0's last frame of the same step
    assert i0[B + 1] == i1[B] and np.array_equal(f0[B + 1], f1[B])         # next step: rank 0's halo = rank 1's last frame
    assert sorted(set(i0) | set(i1)) == list(range(total))


def _calib_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bodyslam_amd.pipeline import share_calibration
    calls = []

    def make():
        calls.append(rank)
        # a rank calibrating on its own could land on the other side of a tolerance: model it with a rank-dependent report
        return {"class_modes": {"qkv": "wmean" if rank == 0 else "full"}, "neck_mode": "full", "attn_mode": "single", "by": rank}
    rep = share_calibration(make)
    with open(os.path.join(out_dir, f"cal_{rank}"), "w") as f:
        f.write(repr((rep, calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_calibration_is_made_by_rank0_and_shared(tmp_path):
    """ZoeDepthEngine.calibrate picks the correction modes from measurements; in a sharded run every rank must run the SAME arithmetic
    (rank r's depth maps join rank r+1's in one sequence), so rank 0's report is broadcast and nobody else calibrates."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_calib_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    reps = [eval(open(tmp_path / f"cal_{r}").read()) for r in range(world)]
    assert reps[0][0] == reps[1][0] and reps[0][0]["by"] == 0
    assert reps[0][1] == [0] and reps[1][1] == []          # only rank 0 made a report
    # without a process group: the report is simply made
    from bodyslam_amd.pipeline import share_calibration
    assert share_calibration(lambda: {"x": 1}) == {"x": 1}


def _calib_fail_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bodyslam_amd.pipeline import share_calibration

    def make():
        raise MemoryError("reference engine does not fit")       # only ever called on rank 0
    try:
        share_calibration(make)
        msg = "no exception"
    except Exception as e:          # noqa: BLE001
        msg = f"{type(e).__name__}: {e}"
    with open(os.path.join(out_dir, f"fail_{rank}"), "w") as f:
        f.write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_calibration_failure_on_rank0_raises_everywhere(tmp_path):
    """rank 0 failing inside calibrate() (out of memory next to the reference engine, a launch error) must not leave the other ranks
    blocked in the broadcast: the failure is broadcast and raised on every rank (round-4 advisor)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_calib_fail_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = (open(tmp_path / f"fail_{r}").read() for r in range(2))
    assert m0.startswith("MemoryError") and "reference engine does not fit" in m0
    assert m1.startswith("RuntimeError") and "reference engine does not fit" in m1


def test_bench_power_sampler_reads_its_own_board(tmp_path, monkeypatch):
    """bench.py's `power` key: the hwmon node under the device's PCI function is the one sampled (a box's sysfs shows every tenant's GPU),
    median / maximum / cap / clock come out in W and MHz, and a host without readable nodes yields None"""
    import glob
    import time
    import bench
    nodes = []
    for i, (pci, watts, mhz) in enumerate((("0000:05:00.0", 300, 150), ("0000:85:00.0", 1300, 2100))):
        dev_dir = tmp_path / "devices" / pci
        hw = dev_dir / "hwmon" / f"hwmon{i}"
        hw.mkdir(parents=True)
        (hw / "name").write_text("amdgpu\n")
        (hw / "power1_average").write_text(f"{watts * 1000000}\n")
        (hw / "freq1_input").write_text(f"{mhz * 1000000}\n")
        (hw / "power1_cap").write_text("1400000000\n")
        card = tmp_path / "drm" / f"card{i}"
        card.mkdir(parents=True)
        (card / "device").symlink_to(dev_dir, target_is_directory=True)
        nodes.append(str(card / "device" / "hwmon" / f"hwmon{i}"))
    monkeypatch.setattr(glob, "glob", lambda pattern: list(nodes) if "hwmon" in pattern else [])
    s = bench.PowerSampler("0000:05:00.0").start()          # the quieter board is ours: matched by PCI function, not by power
    time.sleep(0.2)
    r = s.result()
    assert r["median_w"] == 300.0 and r["max_w"] == 300.0 and r["cap_w"] == 1400.0 and r["sclk_mhz_median"] == 150.0
    assert r["board"] == "the device's PCI function" and r["samples"] >= 4
    s = bench.PowerSampler(None).start()                    # no PCI function known: the board drawing the most, and the line says so
    time.sleep(0.2)
    r = s.result()
    assert r["median_w"] == 1300.0 and "not matched" in r["board"]
    monkeypatch.setattr(glob, "glob", lambda pattern: [])
    assert bench.PowerSampler("0000:05:00.0").start().result() is None
