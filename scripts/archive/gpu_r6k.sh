#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python scripts/chain_timeline.py 65536 > $O/chain_timeline_f16x2.txt 2>&1
grep -v "distinct\|Warn" $O/chain_timeline_f16x2.txt | cut -c1-200 | head -80
