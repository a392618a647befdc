#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5y2; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python scripts/chain_power.py > $O/chain_power.txt 2>&1
grep -v "Warn\|amdgpu.ids" $O/chain_power.txt | tail -8
