#!/bin/bash
# Everything profiles/README.md lists for a round, in one gpurun call:  bash tools/collect_round.sh r02   (from the repo root)
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/tools/collect_profiles.py $TAG > $OUT/collect.log 2>&1
cd $REPO
python3 bench.py --dtype bf16 --precision fast --single-mode --no-pmc-traffic > $OUT/${TAG}_bench_bf16_fast.json 2> $OUT/bf16.err
python3 bench.py --dtype bf16 --precision accurate --single-mode --no-pmc-traffic > $OUT/bf16_accurate_refusal.txt 2>&1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-pmc-traffic 2> $OUT/torchrun.err | grep "^{" > $OUT/${TAG}_bench_torchrun_n1.json   # (RCCL prints a version banner on stdout)
python3 tools/bench_kernels.py --nb 128 2>/dev/null > $OUT/${TAG}_bench_kernels.txt
ls -la $OUT
tail -c 600 $OUT/${TAG}_bench_n1.json
