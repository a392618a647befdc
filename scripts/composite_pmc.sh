#!/bin/bash
# what bounds nudf_composite_fwd at 32768 x 256: VALU / memory-unit activity per launch (rocprofv3 PMC, kernel-trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4cpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/cp_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/cp_$i -o p -- python $R/scripts/composite_bench.py > $O/pmc_$i.log 2>&1
done
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r4cpmc/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/cp_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "composite_fwd" not in n: continue
        agg[(n.split("(")[0][-40:], row.get("Grid_Size", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k, {n: "%.3e" % (sum(v) / len(v)) for n, v in sorted(c.items())})
PY
cat $O/summary.txt; tail -2 $O/pmc_1.log
