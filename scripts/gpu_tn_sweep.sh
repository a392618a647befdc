#!/bin/bash
# sweep of the weight-gradient GEMM's cost table (time of a k-step of a tile with 1..4 live sub-tiles per wave)
out=gpurun_out/tn_sweep.txt
python scripts/tn_group_bench.py > $out 2>&1
for c in "1,2,3,4" "1.5,2,3,4" "2,2.5,3.2,4" "2,3,3.5,4" "2.5,3,3.5,4" "3,3.3,3.7,4" "4,4,4,4"; do
  TN_BENCH_QUICK=1 NUDF_TN_COSTS=$c python scripts/tn_group_bench.py 2>&1 | grep "costs=\|atomics" >> $out
done
grep -v amdgpu.ids $out
