#!/bin/bash
# round 6: the adjoint sweep forms the second-order term from R and DA (NudfChainStep.X3) -- parity, then A/B at the headline
# size and at config 5's shape
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_kernels.py tests/test_gpu_bf16x3.py tests/test_gpu_chain_t16.py tests/test_gpu_mixed16.py tests/test_gpu_train_parity.py tests/test_gpu_graph.py tests/test_gpu_fullsize.py -m gpu -q -s --tb=short -p no:cacheprovider > $O/pytest.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|TRUE relative" $O/pytest.log | cut -c1-700
for v in 0 1 0 1; do
  NUDF_EX_FLY=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_fly$v.json 2>> $O/bench.err
  python - "$O/bench_fly$v.json" "EX_FLY=$v headline" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms  gemm %.3f ms  hbm_gb/step %.2f" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"], r.get("hbm_gb_per_step", 0)))
for k in r["per_kernel"]:
    if k["class"] == "mlp_chain" and ("tangent" in k["kernel"] or "adjoint" in k["kernel"]):
        print("     %-62s n=%d %.1f us  alg %.0f MB  %.2f TB/s" % (k["kernel"], k["launches"], k["us"], k["algorithmic_mb"], k["gbs"] / 1e3))
PY
done
for v in 0 1 0 1; do
  NUDF_EX_FLY=$v timeout 600 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_fly$v.json 2>> $O/bench.err
  python - "$O/bench_cfg5_fly$v.json" "EX_FLY=$v cfg5 mixed16" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  chains %.3f ms  gemm %.3f ms  hbm_gb/step %.2f" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"], r.get("hbm_gb_per_step", 0)))
for k in r["per_kernel"]:
    if k["class"] == "mlp_chain" and ("tangent" in k["kernel"] or "adjoint" in k["kernel"]):
        print("     %-62s n=%d %.1f us  alg %.0f MB  %.2f TB/s" % (k["kernel"], k["launches"], k["us"], k["algorithmic_mb"], k["gbs"] / 1e3))
PY
done
