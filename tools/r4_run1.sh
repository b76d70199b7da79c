cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_ops_gpu.py -k "attention" -x -q 2>&1 | tail -15) > gpurun_out/r4_run1_ops.log 2>&1
(timeout 400 python tools/probes/attn_variants.py 2>&1 | tail -40) > gpurun_out/r4_run1_attn.log 2>&1
(timeout 300 python tools/bench_kernels.py --nb 128 --only f8 --tiles 9,409,12809 --reps 10 2>&1 | grep -v "amdgpu.ids" | tail -40) > gpurun_out/r4_run1_gemm.log 2>&1
(timeout 900 python -m pytest tests/test_zoedepth_gpu.py -k "reference_precision or adversarial" -x -q 2>&1 | tail -25) > gpurun_out/r4_run1_zoe.log 2>&1
cat gpurun_out/r4_run1_ops.log gpurun_out/r4_run1_attn.log gpurun_out/r4_run1_gemm.log gpurun_out/r4_run1_zoe.log
