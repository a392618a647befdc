#!/bin/bash
# round 3, call B: graph-replay tests, the fixed tests of call A, bench with / without the graph at 512 and 256 rays and in
# the 16-bit mode, RCCL capture probe
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_bucket_alias.py tests/test_gpu_raybatch.py "tests/test_gpu_fullsize_parity.py::test_cfg3_garment_geometry_1024_rays_vs_reference" tests/test_gpu_kernels.py -k "graph or bucket or ref_src or garment or upsample_and_merge" -q -s > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 120 python scripts/rccl_capture_probe.py > $O/rccl_capture.log 2>&1
for G in 1 0; do
  timeout 300 python bench.py --graph $G --no-cpu-baseline --no-roofline > $O/bench_512_g$G.json 2>> $O/bench.err
  timeout 300 python bench.py --graph $G --no-cpu-baseline --no-roofline --rays-per-gpu 256 > $O/bench_256_g$G.json 2>> $O/bench.err
  timeout 300 python bench.py --graph $G --no-cpu-baseline --no-roofline --precision mixed16 > $O/bench_512_mixed16_g$G.json 2>> $O/bench.err
done
tail -n 6 $O/pytest.log; cat $O/rccl_capture.log | tail -n 2
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d.get('forward_only',{}).get('ms'))"; done
