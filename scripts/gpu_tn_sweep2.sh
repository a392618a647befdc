#!/bin/bash
# after the interleaved full-tile loop: cost-table sweep (fp32) and workgroup-count sweep of the 16-bit mode
out=gpurun_out/tn_sweep2.txt
: > $out
for c in "2,3,3.5,4" "2.2,3.2,3.7,4" "2.3,3.4,3.8,4" "2.5,3.5,4,4" "2,3,4,4" "2.2,3.3,3.5,4"; do
  TN_BENCH_QUICK=1 NUDF_TN_COSTS=$c python scripts/tn_group_bench.py 2>&1 | grep "costs=" >> $out
done
for b in 512 448 384 320 256; do
  echo "16-bit mode, NUDF_TNG_BLOCKS=$b" >> $out
  NUDF_TNG_BLOCKS=$b TN_BENCH_16=bb python scripts/tn_group_bench.py 2>&1 | grep "packed k-pair" | tail -1 >> $out
done
cat $out
