#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5c; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python scripts/t16_debug.py 19200 > $O/t16_debug.txt 2>&1
tail -15 $O/t16_debug.txt
