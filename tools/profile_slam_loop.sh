#!/bin/bash
# rocprofv3 kernel stats of the reference's whole loop (tools/probes/slam_loop_full.py):  bash tools/profile_slam_loop.sh [frames] [batch]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loopprof -o loop -- python3 $REPO/tools/probes/slam_loop_full.py ${1:-192} ${2:-64} > $OUT/loopprof.log 2>&1
tail -2 $OUT/loopprof.log
for f in $(find $OUT/loopprof -name "*kernel_stats.csv"); do cp $f $OUT/r03_slam_loop_kernel_stats.csv; done
