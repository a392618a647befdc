#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -14 $O/pytest_gpu.log
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_gpu2.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu2.log
tail -3 $O/pytest_gpu2.log
