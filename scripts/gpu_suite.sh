#!/bin/bash
# the whole GPU suite + smoke on the box (generic runner: O=gpurun_out/<tag>)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-suite}; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
grep -E "^FAILED|^ERROR|passed|failed|pytest exit" $O/pytest_gpu.log | cut -c1-300
tail -n 2 $O/smoke.log
