#!/bin/bash
# weight-gradient GEMM: the interleaved full-tile loop and the XCD-aware workgroup order against their switches
# (NUDF_TN_FLAGS 128 / 64), the grouped-GEMM tests, and the bench line
out=gpurun_out/tn_ab.txt
TN_BENCH_AB=1 python scripts/tn_group_bench.py > $out 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_parity.py -m gpu -q 2>&1 | tail -3 >> $out
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_tn_ab.json
python -c "
import json; d = json.load(open('gpurun_out/bench_tn_ab.json')); print('bench ms/step', d['ms_per_step'], 'kernels', d['kernels'])" >> $out
grep -v amdgpu.ids $out
