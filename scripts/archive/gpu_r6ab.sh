#!/bin/bash
# round 6: where the per-tile scale puts the tile's largest seed (2^-6 default build, 2^0, 2^4): fp16 subnormal operands and speed
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6ab; rm -rf $O; mkdir -p $O
cd $R
for v in base ts0 ts4 base ts0 ts4; do
  L=$R/neuraludf_amd/build/libnudf_$v.so; [ $v = base ] && L=$R/neuraludf_amd/libnudf.so
  NUDF_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg --no-forward-only > $O/bench_$v.json 2>> $O/bench.err
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms  gemm %.3f ms" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]))
print("     " + "  ".join("%s %.0f" % (k["kernel"].split("> ")[1].replace(" P=", "@"), k["us"]) for k in r["per_kernel"] if k["class"] == "mlp_chain"))
PY
done
