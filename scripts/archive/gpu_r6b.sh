#!/bin/bash
# round 6, call b: the f16x2 split (three fp16 MFMA products per fp32 product) on the forward-order sweeps -- accuracy against
# float64, the whole GPU suite in the new default, bench A/B against bf16x3 everywhere
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -m gpu -q -s --tb=short -p no:cacheprovider > $O/x3.log 2>&1
grep -E "passed|failed|against float64|largest relative|Error|assert" $O/x3.log | cut -c1-600
for v in 0 grad 1; do
  NUDF_FWD_F16X2=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_fwd$v.json 2>> $O/bench.err
done
NUDF_FWD_F16X2=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_fwd0_again.json 2>> $O/bench.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6b"
for v in ("0", "grad", "1", "0_again"):
    try:
        d = json.loads(open(f"{O}/bench_fwd{v}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print("FWD_F16X2=%s: %.3f ms  windows %s  power %s W %s MHz  chain class %.3f ms  frac %.3f" % (
            v, d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], d["power"].get("avg_w"), d["power"].get("sclk_mhz_avg"),
            d["kernels"]["mlp_chain"]["ms"], r["frac"]))
        for k in r["per_kernel"]:
            if k["class"] == "mlp_chain":
                print("     %-62s n=%d %.1f us  %.0f TF exec" % (k["kernel"], k["launches"], k["us"], k["tflops"]))
        print("   psnr", d.get("psnr_vs_ref"))
    except Exception as e:
        print(v, "ERR", e)
PY
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1
tail -n 25 $O/pytest_gpu.log | cut -c1-400
