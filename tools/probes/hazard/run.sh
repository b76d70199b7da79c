echo "alone:"; tools/probes/hazard/trans_fwd_probe 10
rm -f /tmp/gpu_load_ready
python tools/probes/gpu_churn.py 60 > /dev/null 2>&1 &
CH=$!
for i in $(seq 1 120); do [ -f /tmp/gpu_load_ready ] && break; sleep 1; done
echo "beside gpu_churn.py:"; tools/probes/hazard/trans_fwd_probe 45
wait $CH
