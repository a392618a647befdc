#!/bin/bash
# one gpurun call: full GPU test suite, the default bench line, a rocprofv3 kernel trace of the same command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-forward-only > $O/prof_bench.log 2>&1)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
tail -n 5 $O/pytest_gpu.log; tail -c 600 $O/bench.json
