#!/bin/bash
# L1 / L2 behaviour of the chain launches (weight-fragment fetches): rocprofv3 PMC passes over tests/chain_sweeps.py at 65 536 points
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4pmc; mkdir -p $O
cat > /tmp/run_sweeps.py <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import torch
from chain_sweeps import sweeps
dev = torch.device("cuda:0")
for _ in range(3):
    sweeps(dev, 65536, 0, seed=3)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_TAG_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- python /tmp/run_sweeps.py > $O/pmc_$i.log 2>&1
done
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r4pmc/l2_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "mlp_chain" not in n: continue
        key = (n.split("(")[0][-28:], row.get("Grid_Size", ""), row.get("Dispatch_Id",""))
        agg[(n.split("(")[0][-28:], row.get("Grid_Size",""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k, {n: "%.3e" % (sum(v) / len(v)) for n, v in sorted(c.items())}, "n=%d" % len(next(iter(c.values()))))
PY
cat $O/l2_summary.txt; tail -3 $O/pmc_1.log
