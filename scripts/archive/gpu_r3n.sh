#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O
timeout 600 python -m pytest "tests/test_gpu_fullsize_parity.py::test_bench_inputs_vs_reference_fixture" -q -s > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
grep -E "bench inputs|passed|failed|rc " $O/pytest.log | tail -n 4
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'])
print(d['psnr_vs_ref']['vs_reference_fixture'])
"
