#!/bin/bash
# f16x2 weight-gradient GEMMs (NUDF_TN_F16X2=1, scales from the sweeps' reported maxima): parity, then A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6v; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_bf16x3.py tests/test_gpu_round6.py tests/test_gpu_train_parity.py tests/test_gpu_kernels.py tests/test_gpu_graph.py -m gpu -q -s --tb=short -p no:cacheprovider > $O/pytest.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|TRUE relative|largest relative" $O/pytest.log | cut -c1-600
for v in 0 1 0 1; do
  NUDF_TN_F16X2=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_tn$v.json 2>> $O/bench.err
  python - "$O/bench_tn$v.json" "NUDF_TN_F16X2=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms  gemm %.3f ms" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]) + "  groups " + " / ".join("%.0f" % k["us"] for k in r["per_kernel"] if k["class"] == "gemm_tn"))
p = d.get("psnr_vs_ref") or {}
PY
done
