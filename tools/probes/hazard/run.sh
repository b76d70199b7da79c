# build (needs hipcc) and run one of the probes alone and beside tools/probes/gpu_churn.py:   bash tools/probes/hazard/run.sh sgpr_war_probe | trans_fwd_probe
P=${1:-sgpr_war_probe}
D=tools/probes/hazard
[ -x $D/$P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $D/$P $D/$P.hip 2>/dev/null
echo "alone:"; $D/$P 10
rm -f /tmp/gpu_load_ready
python tools/probes/gpu_churn.py 60 > /dev/null 2>&1 &
CH=$!
for i in $(seq 1 120); do [ -f /tmp/gpu_load_ready ] && break; sleep 1; done
echo "beside gpu_churn.py:"; $D/$P 45
wait $CH
