cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pct in 0 30 50 70 0; do
  echo "== BS_GEMM_STAGGER=$pct" >> gpurun_out/r4_run6_stagger.log
  (BS_GEMM_STAGGER=$pct timeout 300 python tools/bench_kernels.py --nb 128 --only f8 --tiles 9 --reps 10 2>&1 | grep "wmean\] tile9" ) >> gpurun_out/r4_run6_stagger.log 2>&1
done
cat gpurun_out/r4_run6_stagger.log
