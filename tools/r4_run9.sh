cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python tools/bench_kernels.py --nb 128 --only f8 --tiles 9,409,12809 --reps 10 2>&1 | grep "f8 qkv " ) > gpurun_out/r4_run9_qkv.log 2>&1
cat gpurun_out/r4_run9_qkv.log
