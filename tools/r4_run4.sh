cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python tools/probes/reference_taps.py reference outlier 2>&1 | grep -v amdgpu.ids | grep "depth L1\|layer1 \|layer24\|fused3\|bins3\|depth_net" | tail -10) > gpurun_out/r4_run4_taps_ref.log 2>&1
(timeout 600 python tools/probes/reference_taps.py accurate outlier corr 2>&1 | grep -v amdgpu.ids | grep "depth L1\|layer1 \|layer24\|fused3\|bins3\|depth_net" | tail -10) > gpurun_out/r4_run4_taps_acc.log 2>&1
(timeout 600 python tools/probes/reference_taps.py reference none 2>&1 | grep -v amdgpu.ids | grep "depth L1\|layer1 \|layer24\|fused3\|bins3\|depth_net" | tail -10) > gpurun_out/r4_run4_taps_ref_none.log 2>&1
(timeout 300 python tools/probes/attn_corr_precision.py 2>&1 | grep -v amdgpu.ids | grep "all lo\|elements with" | tail -10) > gpurun_out/r4_run4_prec.log 2>&1
(BS_TEST_REPORT_ONLY=1 timeout 300 python -m pytest tests/test_ops_gpu.py -k "attention_table_corr" -q -s 2>&1 | grep "attention_table_corr\|passed\|failed" | tail -20) > gpurun_out/r4_run4_ops.log 2>&1
(RERUNS=50 timeout 400 python tools/probes/attn_variants.py 2>&1 | tail -40) > gpurun_out/r4_run4_attn.log 2>&1
cat gpurun_out/r4_run4_taps_ref.log gpurun_out/r4_run4_taps_acc.log gpurun_out/r4_run4_taps_ref_none.log gpurun_out/r4_run4_prec.log gpurun_out/r4_run4_ops.log gpurun_out/r4_run4_attn.log
