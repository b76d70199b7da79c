#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
python3 tools/bench_kernels.py --nb 128 --only f8 --tiles 9,51209 --variants wmean --reps 10 2>/dev/null | sed "s/^/d8: /" | cut -c1-110
BODYSLAM_HIP_LIB=$PWD/bodyslam_amd/libbodyslam_hip_d4.so python3 tools/bench_kernels.py --nb 128 --only f8 --tiles 9 --variants wmean --reps 10 2>/dev/null | grep -E " o |fc2" | sed "s/^/d4: /" | cut -c1-110
done
