#!/bin/bash
# end-of-round evidence: full GPU test suite, smoke, headline bench (+ rocprofv3 kernel stats of the same command),
# the two secondary workloads, the composite roofline sizes and the HBM calibration.  Output under gpurun_out/final/.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --workload "dtu_shipped_512x114+32" --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_shipped.json 2>> $O/bench.err
timeout 300 python bench.py --workload garment_blend_1024x128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_blend.json 2>> $O/bench.err
timeout 300 python bench.py --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_mixed16.json 2>> $O/bench.err
python scripts/composite_sizes.py > $O/composite_sizes.txt 2>&1
python scripts/hbm_calib.py >> $O/composite_sizes.txt 2>&1
for w in dtu_scan24_512x128 garment_blend_1024x128; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o bench -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-forward-only > $O/prof_$w.log 2>&1)
  find $O/prof_$w -name "*kernel_trace.csv" -delete
done
tail -n 3 $O/pytest_gpu.log; tail -n 1 $O/smoke.log; for f in bench bench_shipped bench_blend bench_mixed16; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,2), 'M rs/s', 'fwd', round(d['forward_only']['ms'],3), d.get('roofline',{}).get('frac'))
"; done; cat $O/composite_sizes.txt
