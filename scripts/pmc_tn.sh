#!/bin/bash
# PMC passes (separate, kernel-trace only) over the weight-gradient GEMM at the UDF adjoint group, M = 65 536:
# MFMA-busy share, instruction mix, fabric traffic.  usage: pmc_tn.sh [NUDF_TN_FLAGS value]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_tn; rm -rf $O; mkdir -p $O
cd /tmp
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  TN_BENCH_PMC=${1:-0} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python $R/scripts/tn_group_bench.py > $O/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
tot = {}
for f in sorted(glob.glob(R + "/gpurun_out/pmc_tn/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        if "gemm_tn" not in row["Kernel_Name"]: continue
        k = (row["Kernel_Name"][:28], row["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for k, v in agg.items():
        tot[k] = v[1] / max(v[0], 1)
        print(k[0], k[1], "per-dispatch", v[1] / max(v[0], 1), "n", v[0])
for (kern, c), v in tot.items():
    if c == "SQ_VALU_MFMA_BUSY_CYCLES" and (kern, "GRBM_GUI_ACTIVE") in tot:
        el = tot[(kern, "GRBM_GUI_ACTIVE")] / 8.0            # 8 XCDs
        print(f"{kern}: elapsed {el:.0f} cycles, MFMA pipe busy per SIMD {v / 1024:.0f} cycles = {v / 1024 / el * 100:.1f} % of elapsed")
PY
