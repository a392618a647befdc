#!/bin/bash
# f16x2 K loop: weight planes requested 1.33 / 1.67 steps ahead (NUDF_X2_EARLY=1, this tree) vs one step ahead (=0), interleaved
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6o; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_round6.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|against float64" | cut -c1-500
for v in late early late early; do
  L=$R/neuraludf_amd/libnudf.so; [ $v = late ] && L=$R/neuraludf_amd/build/libnudf_x2late.so
  NUDF_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_$v.json 2>> $O/bench.err
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"]))
print("     " + "  ".join("%s %.0f" % (k["kernel"].split("> ")[1].replace(" P=", "@"), k["us"]) for k in r["per_kernel"] if k["class"] == "mlp_chain"))
PY
done
