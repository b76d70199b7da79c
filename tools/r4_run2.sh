cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python tools/probes/attn_corr_precision.py 2>&1 | grep -v amdgpu.ids | tail -60) > gpurun_out/r4_run2_prec.log 2>&1
cat gpurun_out/r4_run2_prec.log
