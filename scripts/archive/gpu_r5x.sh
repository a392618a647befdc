#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5x; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tn_plan.py tests/test_gpu_optim.py -q --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest exit $?" >> $O/pytest_new.log
tail -30 $O/pytest_new.log
