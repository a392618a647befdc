#!/bin/bash
# the three 8 192-point UDF evaluations of the up-sampling rounds: 32-point tiles (256 workgroups, the dispatcher's choice) vs 64-point
# tiles (128 workgroups, half the weight stream) -- NUDF_CHAIN_TILE=64 forces 64 everywhere (the large launches use 64 anyway)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6ag; rm -rf $O; mkdir -p $O
cd $R
for v in 0 64 0 64; do
  NUDF_CHAIN_TILE=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg --no-forward-only > $O/bench_$v.json 2>> $O/bench.err
  python - "$O/bench_$v.json" "NUDF_CHAIN_TILE=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"]))
print("     " + "  ".join("%s %.0f" % (k["kernel"].split("> ")[1].replace(" P=", "@"), k["us"]) for k in r["per_kernel"] if k["class"] == "mlp_chain"))
PY
done
