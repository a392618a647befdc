#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5u; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python scripts/host_profile.py 256 > $O/host_profile.txt 2>&1
grep -v Warn $O/host_profile.txt | head -70
