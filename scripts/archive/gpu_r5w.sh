#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5w; rm -rf $O; mkdir -p $O
cd $R
B="--steps 30 --warmup 8 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg"
timeout 300 python bench.py $B > $O/tile_auto.json 2>> $O/bench.err
NUDF_CHAIN_TILE=32 timeout 300 python bench.py $B > $O/tile_32.json 2>> $O/bench.err
NUDF_CHAIN_TILE=64 timeout 300 python bench.py $B > $O/tile_64.json 2>> $O/bench.err
timeout 300 python bench.py $B > $O/tile_auto2.json 2>> $O/bench.err
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get("power") or {}
        pk = [(e["kernel"][17:45], round(e["us"])) for e in d["roofline"]["per_kernel"] if e["class"] == "mlp_chain"]
        print(f.split("/")[-1], round(d["ms_per_step"], 3), round(p.get("avg_w", 0)), round(p.get("sclk_mhz_avg", 0)), pk)
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/bench.err
