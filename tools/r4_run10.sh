cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python tools/probes/accurate_seeds.py 1 2 3 4 5 6 7 8 2>&1 | grep "^seed") > gpurun_out/r4_run10_seeds.log 2>&1
cat gpurun_out/r4_run10_seeds.log
(cd /tmp && TMPDIR=/tmp timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --single-mode --no-pmc-traffic --no-slam-loop --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4_run10_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4_run10_bench.err)
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_run10_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline_conv_stack"]["frac"], d["roofline_conv_stack"]["executed_frac"])
am = d["accurate_modes"]; print(am["class_modes"], am["attn_mode"], am["neck_mode"], am["l1_abs_vs_reference_m"], am["calibration"]["l1_vs_full_m"])
PY
