#!/bin/bash
# SQ counters of the attention kernel (isolated launch, NB = 128), one rocprofv3 --pmc pass per group:  bash tools/pmc_attention.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"; do
  T=$(echo $G | tr ' ' '_')
  rm -rf /tmp/pa_$T
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pa_$T -o p -- python3 $REPO/tools/bench_kernels.py --nb 128 --only attn --reps 3 > /tmp/pa_$T.log 2>&1 || echo "group failed: $G"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in glob.glob("/tmp/pa_*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p, newline="")):
        if "attention_tab2" in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"][:40]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print(f"   {c:28s} {v / n:16.0f} per launch ({n} launches)")
PY
