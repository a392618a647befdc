#!/bin/bash
# round 3, last call: full GPU suite + smoke + the default bench line of the final tree (+ rocprofv3 stats of the same command)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-forward-only > $O/prof_bench.log 2>&1)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -n 3; tail -n 2 $O/smoke.log
python -c "
import json
d=json.load(open('$O/bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
print(d['psnr_vs_ref']['vs_reference_fixture'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
"
