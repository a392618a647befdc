#!/bin/bash
# round-2 reference lines: other workloads and the rays-per-GPU sweep of the headline workload (fp32)
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweeps_r2.txt
: > $O
for w in dtu_shipped_512x114+32 dtu_scan24_1024x256 garment_blend_1024x128; do
  echo "== workload $w" >> $O
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python scripts/bench_kernels_line.py >> $O
done
for r in 256 512 1024 2048 4096 8192; do
  echo "== dtu_scan24 rays per GPU $r" >> $O
  python bench.py --rays-per-gpu $r --no-cpu-baseline --no-forward-only 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('ms/step %.3f  %.2f M ray-samples/s | chain %.1f TF | gemm_tn %.1f TF' % (d['ms_per_step'], d['value']/1e6, k['mlp_chain']['tflops'], k['gemm_tn']['tflops']))
" >> $O
done
cat $O
