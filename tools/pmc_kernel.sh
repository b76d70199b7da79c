#!/bin/bash
# HBM traffic of the kernels of one probe by PMC, as the guide prescribes: separate --pmc passes, gfx950 correction 2*FETCH + WRITE.
#   bash tools/pmc_kernel.sh tools/probes/tsdf_full_size.py 4
SCRIPT=$1; shift
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pk_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pk_$C -o p -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > /tmp/pk_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0, 0.0, 0.0], "WRITE_SIZE": [0, 0.0, 0.0]})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob(f"/tmp/pk_{c}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p, newline="")):
            if r["Counter_Name"] == c:
                a = agg[r["Kernel_Name"]][c]
                a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"][1] + kv[1]["WRITE_SIZE"][1]))[:8]:
    n = max(v["FETCH_SIZE"][0], 1)
    f, w = v["FETCH_SIZE"][1] / n, v["WRITE_SIZE"][1] / max(v["WRITE_SIZE"][0], 1)
    us = v["FETCH_SIZE"][2] / n / 1e3
    b = (2 * f + w) * 1024
    print(f"{k[:70]:70s} n={n:4d} FETCH {f:10.0f} KB WRITE {w:10.0f} KB -> {b / 1e6:8.1f} MB/launch, {us:8.1f} us under the profiler = {b / (us * 1e-6) / 1e12:5.2f} TB/s")
PY
