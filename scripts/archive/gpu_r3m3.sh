#!/bin/bash
# the default bench line + rocprofv3 kernel statistics of the same command on one box (final tree)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m3
mkdir -p $O; rm -rf $O/*
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --graph 0 --no-cpu-baseline > $O/bench_eager.json 2>> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-forward-only > $O/prof_bench.log 2>&1)
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
python - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3m3"
for n in ("bench","bench_eager"):
    d=json.load(open(O+"/"+n+".json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"])
PY
