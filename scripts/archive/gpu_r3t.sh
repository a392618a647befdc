#!/bin/bash
# config-5 shape, 16-bit mode: rocprofv3 kernel statistics of the bench command (which kernels make up the step?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3t
mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-forward-only > $O/prof_bench.log 2>&1)
tail -n 2 $O/prof_bench.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s %6d %9.1f us avg %6.2f%%" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
