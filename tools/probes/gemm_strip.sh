#!/bin/bash
# Probe: tile order of the plain GEMMs (BS_GEMM_STRIP = strip width in N-tiles, 0 = row-major): launch time and L2 fetch / write bytes per launch
cd ${GRAFT_REPO_ROOT:-.}
REPO=$PWD
OUT=$REPO/gpurun_out/gemm_strip
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 0 2 4 6 0 3; do
  BS_GEMM_STRIP=$w python3 $REPO/tools/bench_kernels.py --nb 128 --only f8 --tiles 9 --variants wmean --reps 10 2>/dev/null | sed "s/^/strip $w: /" >> $OUT/times.txt
done
for w in 0 2 4 6; do
 for c in FETCH_SIZE WRITE_SIZE; do
  BS_GEMM_STRIP=$w timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p_${w}_$c -o p -- python3 $REPO/tools/bench_kernels.py --nb 128 --only f8 --tiles 9 --variants wmean --reps 1 > $OUT/p_${w}_$c.log 2>&1
  f=$(find $OUT/p_${w}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $w $c <<'PY' >> $OUT/traffic.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "igemm_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
rows = rows[len(rows) // 2:]          # the timed round
print(f"strip {sys.argv[2]} {sys.argv[3]} KB per launch (qkv, plain16, fc1lo, o, fc1, fc2):", " ".join(f"{float(r['Counter_Value']):.0f}" for r in rows))
PY
  rm -rf $OUT/p_${w}_$c
 done
done
cat $OUT/times.txt | cut -c1-130; cat $OUT/traffic.txt
