#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summary -> gpurun_out/prof/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof && mkdir -p $R/gpurun_out/prof
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-forward-only "$@" > $R/gpurun_out/prof/run.log 2>&1
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 7 steps", tot / 1e6, "per step", tot / 7e6)
for r in rows[:45]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} total_ms {float(r["TotalDurationNs"])/1e6:8.3f} avg_us {float(r["AverageNs"])/1e3:8.1f} {float(r["Percentage"]):5.1f}%')
PY
find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +1M -delete
