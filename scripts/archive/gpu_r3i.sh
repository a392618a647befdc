#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $O
BENCH_ARGS="--graph 0" bash scripts/trace_step_seq.sh > $O/step_seq_eager.txt 2>&1
BENCH_ARGS="--graph 1" bash scripts/trace_step_seq.sh > $O/step_seq_graph.txt 2>&1
tail -n 1 $O/step_seq_eager.txt $O/step_seq_graph.txt
