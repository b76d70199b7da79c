mkdir -p gpurun_out/r06
rm -f /tmp/gpu_load_ready
python tools/probes/gpu_churn.py 760 > gpurun_out/r06/churn.log 2>&1 &
CH=$!
for i in $(seq 1 120); do [ -f /tmp/gpu_load_ready ] && break; sleep 1; done
DIAG=$PWD/bodyslam_amd/libbodyslam_hip_diag.so
MIXED=$PWD/bodyslam_amd/libbodyslam_hip_diag_mixed.so
for k in 1 2 3 4; do
  echo "pass $k, RELEASE library:"; RAW=1 timeout 200 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
  echo "pass $k, RELEASE library:"; RAW=1 timeout 200 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
  echo "pass $k, control (flagged build, form 1):"; BODYSLAM_HIP_LIB=$MIXED BS_LOGBINOM_INTERLEAVED=1 RAW=1 timeout 200 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
  echo "pass $k, current diagnostics build, form 0:"; BODYSLAM_HIP_LIB=$DIAG BS_LOGBINOM_INTERLEAVED=0 RAW=1 timeout 200 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
done
kill $CH 2>/dev/null; wait $CH 2>/dev/null
echo done
