#!/bin/bash
# round 6, last call: the default bench line (cpu baseline included) and the two secondary workloads that carry PMC traffic, on the
# tree whose profiles/ holds this build's traffic files (so that `traffic_stale` is false)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6final; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_mixed16.json 2>> $O/bench_final.err
timeout 300 python bench.py --workload garment_blend_1024x128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_blend.json 2>> $O/bench_final.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6final/"
d = json.loads(open(O + "bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["ms_per_step"], d["window_ms"], d["value"])
print("roofline", {k: r[k] for k in ("bound", "achieved", "peak", "frac", "traffic", "traffic_stale", "hbm_gb_per_step", "bytes_per_core_sample", "executed_flops_per_algorithmic_flop", "class_ms_by_binding_roof") if k in r})
print("vs_mfma", r.get("vs_mfma"), "fp32eq", r.get("fp32_equivalent"), "power", {k: d["power"][k] for k in ("avg_w", "sclk_mhz_avg")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"), d.get("cpu_baseline", {}).get("sample"))
p = d.get("psnr_vs_ref", {}); print("psnr", p.get("value_db"), p.get("max_abs_diff"), p.get("rays_with_identical_samples"), p.get("vs_reference_fixture", {}).get("rays_with_identical_samples"), p.get("vs_reference_fixture", {}).get("value_db"))
print("fp32", {k: v for k, v in d.get("fp32_exact", {}).items() if k in ("ms_per_step", "frac", "achieved")}, "fwd", d["forward_only"]["ms"])
print("kernels", {k: round(v["ms"], 3) for k, v in d["kernels"].items()})
for f in ("bench_cfg5_mixed16", "bench_blend"):
    e = json.loads(open(O + f + ".json").read().strip().splitlines()[-1]); q = e["roofline"]
    print(f, round(e["ms_per_step"], 3), q.get("bound"), round(q.get("frac", 0), 3), q.get("traffic_stale"), round(q.get("hbm_gb_per_step", 0), 2))
PY
