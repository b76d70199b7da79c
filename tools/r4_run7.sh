cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_zoedepth_gpu.py tests/test_bench_config_gpu.py tests/test_tsdf_gpu.py tests/test_rgbd_odometry_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "bf16_reference or config5 or frame_batch or track_block or engine" 2>&1 | tail -15) > gpurun_out/r4_run7_pytest.log 2>&1
cat gpurun_out/r4_run7_pytest.log
grep "bf16 reference\|config 5 sequence" gpurun_out/*report*.txt | tail -4
(cd /tmp && TMPDIR=/tmp timeout 900 python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r4_run7_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4_run7_bench.err)
tail -3 gpurun_out/r4_run7_bench.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_run7_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "depth_l1_vs_oracle_m", "hbm_allocated_gb")})
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "traffic", "algorithmic_bytes_per_launch")})
print("conv", d["roofline_conv_stack"])
print("other", {k: d["other_mode"][k] for k in ("value", "depth_l1_vs_oracle_m")})
print("slam", d["slam_loop"])
am = d["accurate_modes"]; print("modes", am["class_modes"], am["attn_mode"], am["l1_abs_vs_reference_m"], am["warning"])
print("cpu", d["cpu_baseline"])
for k, v in d["kernels"].items():
    if k.startswith("site:"): print(k, v)
PY
