#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|largest relative|assert" | cut -c1-300
