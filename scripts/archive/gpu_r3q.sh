#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_blend.py "tests/test_gpu_fullsize_parity.py::test_cfg3_mix_sampling_and_blending_vs_reference" "tests/test_gpu_fullsize_parity.py::test_cfg3_garment_geometry_1024_rays_vs_reference" tests/test_gpu_kernels.py tests/test_gpu_dist.py -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
for G in 1 0; do timeout 300 python bench.py --workload garment_blend_1024x128 --steps 10 --warmup 3 --no-cpu-baseline --graph $G > $O/bench_blend_g$G.json 2>> $O/bench.err; done
python -c "
import json
for g in (1,0):
    d=json.load(open('$O/bench_blend_g%d.json'%g)); print(g, d['ms_per_step'], d['config']['launch'][:30])
"
