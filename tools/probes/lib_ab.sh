# same-box A / B of two builds of the library through bench.py (BODYSLAM_HIP_LIB): the build in the tree against bodyslam_amd/libbodyslam_hip_prev.so, which the caller
# makes first (e.g. `git archive <commit> bodyslam_amd/csrc include | tar -x -C /tmp/prev && make -C /tmp/prev/bodyslam_amd/csrc` and copies the .so here)
mkdir -p gpurun_out/r06
for k in 1 2; do
  for v in new prev; do
    if [ $v = prev ]; then export BODYSLAM_HIP_LIB=$PWD/bodyslam_amd/libbodyslam_hip_prev.so; else unset BODYSLAM_HIP_LIB; fi
    python bench.py --single-mode --no-cpu-baseline --no-pmc-traffic --no-slam-loop --no-latency-leg --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['frac'], d['roofline_conv_stack']['frac'])"
  done
done
unset BODYSLAM_HIP_LIB
python tools/bench_kernels.py --nb 128 2>/dev/null | grep -E "^(qkv|oproj|fc1 |fc2|f8 qkv  K1024 N3072 \[wmean\]|f8 fc2.*wmean\]|conv 256->256 @192x256)" | grep -E "tile9|tile10" 
