#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2 3; do
python3 tools/bench_kernels.py --nb 128 --only f8 --tiles 9,102409 --variants wmean --reps 10 2>/dev/null | grep -E "fc1  K1024 N4096 gelu pair-out lo8" | cut -c1-110
done
