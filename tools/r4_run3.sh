cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python tools/probes/reference_taps.py reference outlier 2>&1 | grep -v amdgpu.ids | tail -60) > gpurun_out/r4_run3_taps_ref.log 2>&1
(timeout 600 python tools/probes/reference_taps.py accurate outlier corr 2>&1 | grep -v amdgpu.ids | tail -60) > gpurun_out/r4_run3_taps_acc.log 2>&1
cat gpurun_out/r4_run3_taps_ref.log gpurun_out/r4_run3_taps_acc.log
