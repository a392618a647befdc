#!/bin/bash
# round 6, last call: the default bench line (cpu baseline included) on the tree whose profiles/ holds this build's PMC traffic
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6final; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6final/bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["ms_per_step"], d["window_ms"], d["value"], d["dtype"][:60])
print("roofline", {k: r[k] for k in ("bound", "achieved", "peak", "frac", "traffic", "traffic_stale", "hbm_gb_per_step", "bytes_per_core_sample", "executed_flops_per_algorithmic_flop") if k in r})
print("vs_hbm", r.get("vs_hbm"), "power", d.get("power"))
print("cpu", d.get("cpu_baseline"))
print("psnr", d.get("psnr_vs_ref"))
print("fp32", {k: v for k, v in d.get("fp32_exact", {}).items() if k != "kernels"})
print("fwd", d.get("forward_only"))
PY
