#!/bin/bash
# round 3, call J: fused step loss + seed kernels + copy-free replays: parity tests, then the bench with / without graph
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_step_loss.py tests/test_gpu_graph.py tests/test_gpu_train_parity.py tests/test_gpu_kernels.py tests/test_gpu_fullsize_parity.py tests/test_gpu_dist.py -q -x > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
for G in 1 0; do
  timeout 300 python bench.py --graph $G --no-cpu-baseline > $O/bench_g$G.json 2>> $O/bench.err
  timeout 300 python bench.py --graph $G --no-cpu-baseline --no-roofline > $O/bench_g${G}_b.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d.get('forward_only',{}).get('ms'))"; done
BENCH_ARGS="--graph 1" bash scripts/trace_step_seq.sh > $O/step_seq_graph.txt 2>&1
tail -n 1 $O/step_seq_graph.txt
