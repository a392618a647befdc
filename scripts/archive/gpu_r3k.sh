#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_blend.py -q -x > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 300 python bench.py --workload garment_blend_1024x128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_blend.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_blend.json')); print(d['ms_per_step'], d['config']['launch'][:30])"
