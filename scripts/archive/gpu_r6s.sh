#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -m gpu -q --tb=short -p no:cacheprovider -k "beyond_fp16" 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | cut -c1-300
