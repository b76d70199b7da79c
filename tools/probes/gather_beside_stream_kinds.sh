# which launches of a neighbouring stream make the four-byte-gather form of the last launch differ (profiles/r06_reproducibility.txt (8))
export BODYSLAM_HIP_LIB=$PWD/bodyslam_amd/libbodyslam_hip_diag.so BS_LOGBINOM_INTERLEAVED=${FORM:-5}
run() { echo "== neighbour '$1' (not '$2') x $3"; NEIGHBOUR_ONLY=$1 NEIGHBOUR_NOT=$2 NEIGHBOUR_REPEAT=$3 timeout 300 python tools/probes/gather_beside_stream.py ${N:-10000} 2>&1 | grep "relaunches\|neighbour:\|rror"; }
for spec in "$@"; do IFS=: read a b c <<< "$spec"; run "$a" "$b" "${c:-8}"; done
