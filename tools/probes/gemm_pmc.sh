#!/bin/bash
# Probe: stall / instruction-mix counters of the backbone GEMM launches (tools/bench_kernels.py, f8 "wmean" shapes, tile 9, one rep),
# one rocprofv3 --pmc pass per counter group.  Output: gpurun_out/gemm_pmc/<group>.csv (one row per dispatch and counter).
cd ${GRAFT_REPO_ROOT:-.}
REPO=$PWD
OUT=$REPO/gpurun_out/gemm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 $REPO/tools/bench_kernels.py --nb 128 --only f8 --tiles 9,409 --variants wmean --reps 1 > $OUT/g$i.log 2>&1
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$grp" <<'PY' > $OUT/g$i.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if "igemm_kernel" not in r["Kernel_Name"]:
        continue
    key = (int(r["Dispatch_Id"]), int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("# dispatch blocks " + sys.argv[2])
for (d, g), c in by.items():
    print(d, g, " ".join(f"{k}={v:.4g}" for k, v in c.items()))
PY
  rm -rf $OUT/g$i
done
ls $OUT; cat $OUT/g1.txt | tail -30
