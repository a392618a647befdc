#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O; rm -f $O/probe.txt
run() { timeout 120 python scripts/graph_probe.py "$@" > $O/p.log 2>&1; echo "rc=$? args=$*  $(grep -E '^OK|Error|error' $O/p.log | tail -n 1)" >> $O/probe.txt; }
run 512 64 64 4 0 dtu 0
run 192 32 32 2 1 tiny 1
cat $O/probe.txt
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_bucket_alias.py tests/test_gpu_raybatch.py "tests/test_gpu_fullsize_parity.py::test_cfg3_garment_geometry_1024_rays_vs_reference" tests/test_gpu_kernels.py -k "graph or bucket or ref_src or garment or upsample_and_merge" -q -s > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |cfg3 mix|Error|assert" $O/pytest.log | tail -n 12
