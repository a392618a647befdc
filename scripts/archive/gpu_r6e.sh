#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for v in 1 0; do for no in 0 8; do
NUDF_FWD_F16X2=$v timeout 600 python scripts/debug_train_parity.py $no 2>&1 | grep -E "precision|dw|movement" | head -12
done; done
NUDF_PRECISION=fp32 timeout 600 python scripts/debug_train_parity.py 8 2>&1 | grep -E "precision|dw|movement" | head -12
