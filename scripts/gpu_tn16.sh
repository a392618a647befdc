#!/bin/bash
# weight-gradient GEMM, 16-bit MFMA mode: the packed k-pair LDS image against the generic kernel (NUDF_TN_FLAGS bit 256)
out=gpurun_out/tn16.txt
: > $out
for k in bb ff bf; do
  echo "== operand kinds $k, M = 65536" >> $out
  TN_BENCH_16=$k python scripts/tn_group_bench.py >> $out 2>&1
done
echo "== operand kinds bb, M = 262144" >> $out
TN_BENCH_16=bb python scripts/tn_group_bench.py 262144 >> $out 2>&1
echo "== operand kinds bb, M = 1000 (ragged)" >> $out
TN_BENCH_16=bb python scripts/tn_group_bench.py 1000 >> $out 2>&1
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm_tn" 2>&1 | tail -3 >> $out
grep -v amdgpu.ids $out
