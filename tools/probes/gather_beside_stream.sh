export BODYSLAM_HIP_LIB=$PWD/bodyslam_amd/libbodyslam_hip_diag.so
for m in 5 0; do echo "form $m:"; BS_LOGBINOM_INTERLEAVED=$m timeout 600 python tools/probes/gather_beside_stream.py 20000 2>&1 | grep "relaunches\|Error\|error" ; done
