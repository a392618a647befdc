#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python scripts/debug_train_parity.py 8 2>&1 | grep -v Warn | tail -40
NUDF_FWD_F16X2=0 timeout 600 python scripts/debug_train_parity.py 8 2>&1 | grep -v Warn | tail -40
NUDF_FWD_F16X2=0 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_parity.py tests/test_gpu_blend.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed|Error" | cut -c1-300
