#!/bin/bash
# round 6, call a: the fixture tests with TRUE relative gradient bars (tests/common.py: param_grads_vs_reference), once on the
# round-5 library and once on this tree's (composite backward: cancellation-free logistic derivative); baseline bench of the box.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a; rm -rf $O; mkdir -p $O
cd $R
NUDF_LIB=$R/neuraludf_amd/libnudf_r5.so timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -s --tb=short -p no:cacheprovider -k "vs_reference" > $O/grads_r5lib.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -s --tb=short -p no:cacheprovider -k "vs_reference" > $O/grads_new.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|relative|AssertionError|assert " $O/grads_r5lib.log | cut -c1-900 | tail -30
echo ---- new
grep -E "passed|failed|relative|AssertionError|assert " $O/grads_new.log | cut -c1-900 | tail -30
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('window_ms'), d.get('power'))"
