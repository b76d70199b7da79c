"""Helper of tests/test_pipeline_gpu.py::test_two_process_sharded_sequence (not a test module): SURVEY 8(e)'s sharded path with a REAL
process group on the GPU box.  Two fresh processes share the one GPU of the box over a `gloo` group (RCCL needs one device per rank) and
run ``BodySlamPipeline.run_sequence(frames, rank, 2)`` with NO gather hook: the load-time calibration goes through
``share_calibration`` (all-reduce + object broadcast), the relative poses through ``gather_relative_poses`` (staged through the host
for gloo).  A third fresh process runs the sequence unsharded.  Everything is written to <outdir>/*.npz; the test compares.
tests/conftest.py starts this script at session start, BEFORE the pytest process touches the GPU.

    python tests/two_process_shard.py <outdir>"""
import dataclasses
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FRAMES, H, W, BATCH, TARGET = 7, 160, 192, 2, (64, 96)


def _case():
    import numpy as np  # noqa: F401
    from bodyslam_amd.synthetic import make_sequence
    from bodyslam_amd.zoedepth import ZoeConfig
    from oracle import cyclepose_ref as CP
    from oracle import zoedepth_ref as Z
    cfg_o = Z.ZoeConfig(hidden=128, layers=4, heads=2, intermediate=256, taps=(1, 2, 3, 4), image_size=64)
    names = {f.name for f in dataclasses.fields(ZoeConfig)}
    cfg_p = ZoeConfig(**{k: v for k, v in dataclasses.asdict(cfg_o).items() if k in names})
    return cfg_p, Z.synth_weights(cfg_o, seed=6), CP.synth_weights(seed=6), make_sequence(N_FRAMES, H, W, seed=21)


def _worker(rank, world, port, outdir):
    import numpy as np
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    from bodyslam_amd.pipeline import BodySlamPipeline
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=300))
    cfg_p, wz, wp, frames = _case()
    pipe = BodySlamPipeline(wz, wp, cfg_p, batch=BATCH, target_hw=TARGET, precision="accurate")
    res = pipe.run_sequence(frames, rank, world, keep_depth_m=True)
    cal = pipe.zoe.calibration or {}
    np.savez(os.path.join(outdir, f"w{world}_r{rank}.npz"), start=res.start, end=res.end, depth_u16=res.depth_u16.cpu().numpy(),
             depth_m=res.depth_m.cpu().numpy(), t_rel=res.t_rel.cpu().numpy(), g_abs=res.g_abs.cpu().numpy(),
             counts=res.point_counts.cpu().numpy(), modes=repr((cal.get("class_modes"), cal.get("neck_mode"), cal.get("attn_mode"))))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(outdir):
    import torch.multiprocessing as mp
    os.makedirs(outdir, exist_ok=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, outdir), nprocs=2, join=True)
    mp.spawn(_worker, args=(1, port, outdir), nprocs=1, join=True)
    with open(os.path.join(outdir, "done"), "w") as f:
        f.write("ok")


if __name__ == "__main__":
    main(sys.argv[1])
