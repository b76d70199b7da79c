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
python3 bench.py --dtype bf16 --precision reference --no-pmc-traffic --no-slam-loop > $OUT/${TAG}_bench_bf16_reference.json 2> $OUT/bf16_ref.err
# BASELINE config 5's frame size on one GPU (1280x1024 -> a 416x512 network input, 833 tokens), 16 frames per step
python3 bench.py --height 1024 --width 1280 --batch 16 --no-pmc-traffic --no-slam-loop > $OUT/${TAG}_bench_1280x1024_n1.json 2> $OUT/cfg5.err
python3 tools/probes/plan_call_times.py > $OUT/${TAG}_plan_call_times.txt 2>/dev/null
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-pmc-traffic 2> $OUT/torchrun.err | grep "^{" > $OUT/${TAG}_bench_torchrun_n1.json   # (RCCL prints a version banner on stdout)
python3 tools/bench_kernels.py --nb 128 2>/dev/null > $OUT/${TAG}_bench_kernels.txt
# round 5: the driver's command line, the accurate mode on the adversarial weight sets, the per-site neck study, the fused upsample + conv2, eight seeds
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_like_n1.json 2> $OUT/driver_like.err
python3 bench.py --weights outlier --single-mode --no-pmc-traffic --no-slam-loop > $OUT/${TAG}_bench_outlier_weights.json 2> $OUT/outlier.err
python3 tools/probes/neck_site_study.py 0 5 > $OUT/${TAG}_neck_site_study.txt 2>/dev/null
python3 tools/probes/neck_plain_study.py 0 2>/dev/null > $OUT/${TAG}_neck_plain_study.txt
python3 tools/probes/upconv_fused_time.py > $OUT/${TAG}_upconv_fused.txt 2>/dev/null
ABLATE=1 python3 tools/probes/upconv_fused_time.py >> $OUT/${TAG}_upconv_fused.txt 2>/dev/null
python3 tools/probes/attn_ablate.py > $OUT/${TAG}_attention_ablations.txt 2>/dev/null
python3 tools/probes/power_trace.py > $OUT/${TAG}_power_trace.txt 2>/dev/null
python3 tools/probes/accurate_seeds.py 1 2 3 4 5 6 7 8 2>/dev/null | cut -c1-400 > $OUT/${TAG}_accurate_seeds.txt
ls -la $OUT
tail -c 600 $OUT/${TAG}_bench_n1.json
