#!/bin/bash
# round 3, call A: full GPU test suite (incl. the new garment / mixed16-cfg5 / bucket-alias / localisation cases) and the
# default bench line with the new per_kernel roofline + reference-kind CPU baseline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
cp gpurun_out/parity_localisation.txt $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -n 5; tail -c 1500 $O/bench.json
