#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -s --tb=short -p no:cacheprovider -k cfg5_shape > $O/cfg5.log 2>&1
grep -E "passed|failed|cfg5 shape|Error|assert" $O/cfg5.log | cut -c1-1200
NUDF_FWD_F16X2=0 timeout 600 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -s --tb=short -p no:cacheprovider -k cfg5_shape > $O/cfg5_x3.log 2>&1
grep -E "passed|failed|cfg5 shape|Error|assert" $O/cfg5_x3.log | cut -c1-1200
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu.log | cut -c1-300
