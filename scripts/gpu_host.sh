#!/bin/bash
# eager host cost A/B on one box: NUDF_HOST_FAST=0 (round-5 host path) vs 1 (round 6), 256 and 512 rays, interleaved twice
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-host}; mkdir -p $O
cd $R
for rep in 1 2; do for rays in 256 512; do for fast in 0 1; do
  echo -n "rays $rays NUDF_HOST_FAST=$fast: "; NUDF_HOST_FAST=$fast timeout 600 python scripts/host_profile.py $rays 2>&1 | grep "host enqueue"
done; done; done | tee $O/host_ab.txt
NUDF_HOST_FAST=1 timeout 600 python scripts/host_profile.py 256 --profile 2>&1 | grep -E "tottime|mlp.py|_lib.py|optim.py|run_backward|empty|train.py|blending.py|loss.py|dist.py" | head -24 | cut -c1-160 | tee $O/host_profile_top.txt
for g in 0 1; do timeout 600 python bench.py --graph $g --no-cpu-baseline --no-fp32-leg --no-roofline > $O/bench_graph$g.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bench_graph$g.json').read().strip().splitlines()[-1]); print('bench --graph $g', d['ms_per_step'], d['window_ms'])"; done
