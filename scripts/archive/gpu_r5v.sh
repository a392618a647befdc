#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5v2; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_optim.py tests/test_gpu_graph.py tests/test_gpu_train_parity.py -q --tb=short -p no:cacheprovider -x > $O/pytest_new.log 2>&1; echo "pytest exit $?" >> $O/pytest_new.log
grep -E "passed|failed|error" $O/pytest_new.log | tail -3
timeout 300 python scripts/host_profile.py 256 > $O/host_profile.txt 2>&1
grep -v Warn $O/host_profile.txt | head -16
