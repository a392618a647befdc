#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5o; rm -rf $O; mkdir -p $O
cd $R
NUDF_LIB=$R/neuraludf_amd/build/libnudf_stamps.so timeout 300 python scripts/tn3_phases.py 65536 > $O/tn3_phases.txt 2>&1
timeout 300 python scripts/tn3_phases.py 65536 2>&1 | head -3 > $O/tn3_ship.txt
cat $O/tn3_phases.txt | grep -v Warn
cat $O/tn3_ship.txt | grep "per launch"
