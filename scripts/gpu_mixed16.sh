#!/bin/bash
# 16-bit mode after the packed-image weight-gradient kernel: tests, the two mixed16 bench lines
out=gpurun_out/mixed16.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mixed16.py -m gpu -q -k "gemm_tn or mixed16 or 16" 2>&1 | tail -4 > $out
python bench.py --precision mixed16 2>/dev/null | tail -1 > gpurun_out/bench_mixed16.json
python bench.py --precision mixed16 --workload dtu_scan24_1024x256 2>/dev/null | tail -1 > gpurun_out/bench_cfg5_mixed16.json
python -c "
import json
for f in ('gpurun_out/bench_mixed16.json', 'gpurun_out/bench_cfg5_mixed16.json'):
    d = json.load(open(f)); print(f, 'ms/step', round(d['ms_per_step'], 3), {k: (round(v['ms'], 3), round(v['tflops'], 1)) for k, v in d['kernels'].items()}, 'psnr', d.get('psnr_vs_ref', {}).get('value_db'))" >> $out
cat $out
