#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_graph.py -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -n 15 $O/pytest.log
