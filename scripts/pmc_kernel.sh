#!/bin/bash
# usage: pmc_kernel.sh <kernel-substring> -- <command...>   : PMC passes (separate) over one kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift; shift
OUT=$R/gpurun_out/pmc_$K
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- "$@" > $OUT/p$i.log 2>&1
done
python - "$K" "$OUT" <<'PY'
import csv, glob, os, sys, collections
K, OUT = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(OUT + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        if K not in row["Kernel_Name"]: continue
        k = (row["Kernel_Name"][:48], row["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for k, v in agg.items():
        print(k[0], k[1], "per-dispatch", v[1] / max(v[0], 1), "n", v[0])
PY
