#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6j; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_chain_t16.py tests/test_gpu_mixed16.py tests/test_gpu_fullsize_parity.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error" | cut -c1-300
for v in 1 2; do
  timeout 600 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_$v.json 2>> $O/bench.err
  python - "$O/bench_cfg5_$v.json" "cfg5 mixed16 run $v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  chains %.3f ms  gemm %.3f ms" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]))
for k in r["per_kernel"][:11]:
    print("     %-62s n=%d %.1f us  alg %.0f MB  %.2f TB/s" % (k["kernel"], k["launches"], k["us"], k["algorithmic_mb"], k["gbs"] / 1e3))
PY
done
