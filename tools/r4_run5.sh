cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python tools/probes/reference_taps.py reference outlier 2>&1 | grep -v amdgpu.ids | grep "depth L1\|layer1 \|layer24\|fused3\|bins3\|depth_net" | tail -10) > gpurun_out/r4_run5_taps_ref.log 2>&1
(timeout 600 python tools/probes/reference_taps.py accurate outlier corr 2>&1 | grep -v amdgpu.ids | grep "depth L1\|layer1 \|layer24\|fused3\|bins3\|depth_net" | tail -10) > gpurun_out/r4_run5_taps_acc.log 2>&1
cat gpurun_out/r4_run5_taps_ref.log gpurun_out/r4_run5_taps_acc.log
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > gpurun_out/r4_run5_pytest.log 2>&1
cat gpurun_out/r4_run5_pytest.log
