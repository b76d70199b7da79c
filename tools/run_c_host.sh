#!/bin/bash
# A host without Python runs the depth network: export an engine (python), run examples/zoedepth_host.c on raw frames (a plain C program
# linked against libbodyslam_hip.so only), compare with the Python-driven plan (python).  Three separate processes, in sequence.
#   bash tools/run_c_host.sh            (from the repo root, on a GPU box)
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c_host
mkdir -p $OUT
cd $REPO
gcc -O2 -Wall -Iinclude examples/zoedepth_host.c -o $OUT/zoedepth_host -Lbodyslam_amd -lbodyslam_hip -Wl,-rpath,$REPO/bodyslam_amd
python3 - <<PY
import os, sys, time, torch
sys.path.insert(0, "$REPO")
from bodyslam_amd.engine_export import export_zoedepth
from bodyslam_amd.synthetic import make_sequence, random_zoedepth_weights
from bodyslam_amd.zoedepth import ZoeConfig, ZoeDepthEngine
cfg = ZoeConfig()                                            # the full ZoeD_NK (BEiT-L), accurate mode
eng = ZoeDepthEngine(random_zoedepth_weights(cfg, seed=0), cfg, precision="accurate")
B, H, W = 2, 480, 640
t0 = time.time()
info = export_zoedepth(eng, B, H, W, "$OUT/zoed_nk_b2.bseng")
print("export:", info, f"{time.time() - t0:.1f} s, file {os.path.getsize('$OUT/zoed_nk_b2.bseng') / 1e9:.2f} GB, class modes {eng.class_modes}")
frames = torch.from_numpy(make_sequence(B, H, W, seed=5))
frames.numpy().tofile("$OUT/frames.u8")
dm, _ = eng.infer(frames.cuda())
dm.cpu().numpy().tofile("$OUT/depth_python.f32")
PY
$OUT/zoedepth_host $OUT/zoed_nk_b2.bseng $OUT/frames.u8 2 480 640 $OUT/depth_c.f32
python3 - <<PY
import numpy as np
a = np.fromfile("$OUT/depth_python.f32", dtype=np.float32); b = np.fromfile("$OUT/depth_c.f32", dtype=np.float32)
print(f"C host vs Python-driven plan: {a.size} depth values, identical: {bool(np.array_equal(a, b))}, max |diff| {np.abs(a - b).max():.3e}, depth range {a.min():.3f}..{a.max():.3f} m")
PY
rm -f $OUT/zoed_nk_b2.bseng $OUT/frames.u8 $OUT/depth_python.f32 $OUT/depth_c.f32 $OUT/zoedepth_host
