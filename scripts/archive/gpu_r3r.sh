#!/bin/bash
# per-wave timelines of the 16-bit chain sweeps at the config-5 point count (where do the cycles go: K loops or epilogues?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $O
NUDF_PRECISION=mixed16 timeout 300 python scripts/chain_timeline.py 262144 > $O/timeline_udf_mixed16.txt 2>&1
NUDF_PRECISION=mixed16 TIMELINE_COLOUR=1 timeout 300 python scripts/chain_timeline.py 262144 > $O/timeline_colour_mixed16.txt 2>&1
grep -v "distinct\|simd" $O/timeline_udf_mixed16.txt | grep "launch\|per wave\|step  1\|step  7"
