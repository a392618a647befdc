#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_abi.py -m "gpu or not gpu" -q -s --tb=short -p no:cacheprovider -k "f16x2_weight or abi or layout" 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|error vs|assert" | cut -c1-400
