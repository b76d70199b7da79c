#!/bin/bash
# kernel-trace summary of one bench step (accurate mode unless $1 says otherwise): top kernels by total time
MODE=${1:-accurate}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o pb -- python3 $GRAFT_REPO_ROOT/bench.py --single-mode --precision $MODE --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-pmc-traffic > /tmp/pb.log 2>&1
f=$(find /tmp/pb -name 'pb_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:110]:110s} n={int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.2f} %')
PY
