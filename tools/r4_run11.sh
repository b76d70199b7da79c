cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python tools/bench_kernels.py --nb 128 --only f8 --tiles 9,51209,102409,9 --reps 10 2>&1 | grep "f8 qkv  K1024 N3072 \[wmean\]" ) > gpurun_out/r4_run11_qkv.log 2>&1
cat gpurun_out/r4_run11_qkv.log
