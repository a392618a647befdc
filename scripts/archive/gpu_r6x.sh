#!/bin/bash
# round 6: f16x2 weight-gradient GEMM with a double-buffered LDS image (one barrier per k-step) against the single image (tn2sb)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6x; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_bf16x3.py tests/test_gpu_fullsize_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|pytest exit" $O/pytest.log | cut -c1-300
for v in base tn2sb base tn2sb; do
  L=$R/neuraludf_amd/build/libnudf_$v.so; [ $v = base ] && L=$R/neuraludf_amd/libnudf.so
  NUDF_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg --no-forward-only > $O/bench_$v.json 2>> $O/bench.err
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms  gemm %.3f ms" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]))
print("     groups " + " / ".join("%.0f" % k["us"] for k in r["per_kernel"] if k["class"] == "gemm_tn"))
PY
done
