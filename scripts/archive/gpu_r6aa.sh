#!/bin/bash
# round 6: f16x2 backward sweeps with per-tile scaling (NudfChain.tile_scale): unit tests, parity suites, A/B against NUDF_BWD_F16X2=0
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6aa; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_bf16x3.py tests/test_gpu_fullsize_parity.py tests/test_gpu_kernels.py tests/test_gpu_train_parity.py tests/test_gpu_graph.py -m gpu -q --tb=short -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|pytest exit|largest relative" $O/pytest.log | cut -c1-300
for v in 0 1 0 1; do
  NUDF_BWD_F16X2=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg --no-forward-only > $O/bench_bwd$v.json 2>> $O/bench.err
  python - "$O/bench_bwd$v.json" "NUDF_BWD_F16X2=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms  gemm %.3f ms" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]))
print("     " + "  ".join("%s %.0f" % (k["kernel"].split("> ")[1].replace(" P=", "@"), k["us"]) for k in r["per_kernel"] if k["class"] == "mlp_chain"))
PY
done
tail -3 $O/bench.err
