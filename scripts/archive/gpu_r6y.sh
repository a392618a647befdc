#!/bin/bash
# TIMING PROBE: tangent / adjoint / ReLU-backward sweeps on three fp16 products (f16x2) with the loss seed scaled by 2^24 so that the
# adjoints lie in fp16's range (Adam is scale-invariant, so the step trains on sensible weights) -- what a range-safe f16x2
# backward could be worth
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6y; rm -rf $O; mkdir -p $O
cd $R
for v in base probe base probe; do
  if [ $v = probe ]; then export NUDF_PROBE_BWD_F16X2=1; else unset NUDF_PROBE_BWD_F16X2; fi
  NUDF_PROBE_SEED_LOG2=24 timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg --no-forward-only > $O/bench_$v.json 2>> $O/bench.err
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f ms  windows %s  power %s W %s MHz  chains %.3f ms  gemm %.3f ms  psnr %s" % (sys.argv[2], d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], round(d["power"].get("avg_w", 0)), round(d["power"].get("sclk_mhz_avg", 0)), d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"], d.get("psnr_vs_ref", {}).get("value_db")))
print("     " + "  ".join("%s %.0f" % (k["kernel"].split("> ")[1].replace(" P=", "@"), k["us"]) for k in r["per_kernel"] if k["class"] == "mlp_chain"))
PY
done
tail -3 $O/bench.err
