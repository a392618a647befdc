#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5y; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python scripts/overlap_probe.py > $O/overlap.txt 2>&1
grep -v Warn $O/overlap.txt | tail -8
