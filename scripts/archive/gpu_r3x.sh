#!/bin/bash
# after the 16-bit work: full GPU suite, fresh PMC traffic of the 16-bit chains at the config-5 shape, the two mixed16 bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3x
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -n 4 $O/pytest_gpu.log
bash scripts/pmc_traffic.sh mlp_chain -- python $R/bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-forward-only --graph 0 > $O/traffic_mlp_chain_cfg5_mixed16.json 2> $O/traffic.err
cp $O/traffic_mlp_chain_cfg5_mixed16.json $R/profiles/r03_traffic_mlp_chain_cfg5_mixed16.json
timeout 600 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 20 --warmup 5 > $O/bench_cfg5_1024x256_mixed16.json 2> $O/bench1.err
timeout 600 python bench.py --precision mixed16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_mixed16.json 2> $O/bench2.err
python - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3x"
for n in ("bench_cfg5_1024x256_mixed16","bench_mixed16"):
    try:
        d=json.load(open(O+"/"+n+".json"))
        print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("vs_hbm"), d.get("psnr_vs_ref"))
    except Exception as e: print(n, "ERR", e)
PY
