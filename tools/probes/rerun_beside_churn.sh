# the last launch's hidden layer in its round-6 form and in the rounds 2-5 form (diagnostics build, BS_LOGBINOM_INTERLEAVED=1), rerun beside
# tools/probes/gpu_churn.py:   bash tools/probes/rerun_beside_churn.sh   (profiles/r06_reproducibility.txt (6))
mkdir -p gpurun_out/r06
rm -f /tmp/gpu_load_ready
python tools/probes/gpu_churn.py ${CHURN_S:-330} > gpurun_out/r06/churn.log 2>&1 &
CH=$!
for i in $(seq 1 120); do [ -f /tmp/gpu_load_ready ] && break; sleep 1; done
export BODYSLAM_HIP_LIB=$PWD/bodyslam_amd/libbodyslam_hip_diag.so
for k in 1 2; do
  echo "pass $k, fenced LDS reads (the shipped form):"; RAW=1 DBG=1 timeout 200 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
  echo "pass $k, LDS reads interleaved with the scalar loads (rounds 2-5):"; BS_LOGBINOM_INTERLEAVED=1 RAW=1 DBG=1 timeout 200 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
done
wait $CH
tail -1 gpurun_out/r06/churn.log
