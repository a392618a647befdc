#!/bin/bash
# one GPU-box round: parity tests, smoke, bench, rocprof kernel trace.  Output under gpurun_out/.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m neuraludf_amd.build > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -n 60 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -n 3 gpurun_out/bench.log
if [ "${NUDF_PROFILE:-1}" = "1" ]; then
  rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-forward-only > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1)
  find gpurun_out/prof -name "*kernel_stats*" | head -3
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
  find gpurun_out/prof -name "*kernel_trace.csv" -size +1M -delete
fi
