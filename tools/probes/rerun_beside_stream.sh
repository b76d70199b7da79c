# every marked stage of the small plan, rerun beside a full-size plan of the SAME process on a second stream (profiles/r06_reproducibility.txt (8)):
# the release library, and the diagnostics build's four-byte-gather form of the last launch as the control
export NEIGHBOUR_STREAM=1 RAW=1
for k in 1 2; do
  echo "pass $k, RELEASE library:"; timeout 400 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
  echo "pass $k, control (diagnostics build, four-byte gathers in the last launch):"; BODYSLAM_HIP_LIB=$PWD/bodyslam_amd/libbodyslam_hip_diag.so BS_LOGBINOM_INTERLEAVED=5 timeout 400 python tools/probes/rerun_determinism.py 400 2>&1 | grep "reruns of"
done
