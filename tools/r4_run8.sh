cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python tools/bench_kernels.py --nb 128 --only f8 --tiles 9,25609 --reps 10 2>&1 | grep "f8 qkv " ) > gpurun_out/r4_run8_qkv.log 2>&1
cat gpurun_out/r4_run8_qkv.log
(timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_bench_config_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "qkv or attention_table or bench_batch or engine or gemm_f8" 2>&1 | tail -6) > gpurun_out/r4_run8_pytest.log 2>&1
cat gpurun_out/r4_run8_pytest.log
