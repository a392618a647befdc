#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $O
TIMELINE_COLOUR=1 timeout 300 python scripts/chain_timeline.py 65536 > $O/timeline_colour.txt 2>&1
timeout 300 python scripts/chain_timeline.py 8192 > $O/timeline_8192.txt 2>&1
grep -v Warning $O/timeline_colour.txt | tail -n 60
grep -v Warning $O/timeline_8192.txt | head -n 22
