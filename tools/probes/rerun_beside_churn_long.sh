# the last launch's hidden layer in its forms (diagnostics builds), rerun beside tools/probes/gpu_churn.py (profiles/r06_reproducibility.txt (6)).
#   form 0: shipped -- eight units' corners in one go between two lgkmcnt(0) fences
#   form 4: the rounds 2-5 reads (two units' corners per pass, inside the loop) with a fence in front of and behind them
#   form 1: the rounds 2-5 reads among the scalar loads.  Whether the compiler really places them there depends on the surrounding code: the control
#           arm uses the diagnostics library built from commit fb3c4d5, whose ISA tools/probes/lgkm_mix_audit.py flags (MIXED_LIB: `git archive fb3c4d5
#           bodyslam_amd/csrc include | tar -x -C /tmp/x && make -C /tmp/x/bodyslam_amd/csrc DIAG=1`), not the current diagnostics build
mkdir -p gpurun_out/r06
rm -f /tmp/gpu_load_ready
python tools/probes/gpu_churn.py ${CHURN_S:-760} > gpurun_out/r06/churn.log 2>&1 &
CH=$!
for i in $(seq 1 120); do [ -f /tmp/gpu_load_ready ] && break; sleep 1; done
DIAG=$PWD/bodyslam_amd/libbodyslam_hip_diag.so
MIXED=${MIXED_LIB:-$PWD/bodyslam_amd/libbodyslam_hip_diag_mixed.so}
for k in $(seq 1 ${PASSES:-4}); do
  for m in ${MODES:-0 0 1}; do
    lib=$DIAG; [ $m = 1 ] && lib=$MIXED
    echo "pass $k, form $m:"; BODYSLAM_HIP_LIB=$lib BS_LOGBINOM_INTERLEAVED=$m RAW=1 timeout 200 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
  done
done
kill $CH 2>/dev/null; wait $CH 2>/dev/null
echo done
