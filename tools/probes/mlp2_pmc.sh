#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
REPO=$PWD
python3 tools/probes/mlp2_time.py
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/mp$i -o p -- python3 $REPO/tools/probes/mlp2_time.py 2 > /tmp/mp$i.log 2>&1
  f=$(find /tmp/mp$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
by = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "mlp2" in r["Kernel_Name"]:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
d = list(by.values())[-1]
print(" ".join(f"{k}={v:.4g}" for k, v in d.items()))
PY
done
