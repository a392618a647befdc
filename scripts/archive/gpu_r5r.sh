#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5r; rm -rf $O; mkdir -p $O
cd $R
NUDF_LIB=$R/neuraludf_amd/build/libnudf_stamps.so timeout 300 python scripts/tn3_power.py > $O/tn3_power.txt 2>&1
grep -v Warn $O/tn3_power.txt | tail -60
