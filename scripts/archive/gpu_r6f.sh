#!/bin/bash
# probe: what the f16x2 sweeps cost when the K loop does not split (upper bound of a tile that holds the two fp16 parts)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_base.json 2>> $O/bench.err
NUDF_LIB=$R/neuraludf_amd/build/libnudf_x2nosplit.so timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_nosplit.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_base2.json 2>> $O/bench.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6f"
for v in ("base", "nosplit", "base2"):
    try:
        d = json.loads(open(f"{O}/bench_{v}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%s: %.3f ms  windows %s  power %s W %s MHz  chain class %.3f ms" % (
            v, d["ms_per_step"], [round(w, 3) for w in d["window_ms"]], d["power"].get("avg_w"), d["power"].get("sclk_mhz_avg"),
            d["kernels"]["mlp_chain"]["ms"]))
        for k in r["per_kernel"]:
            if k["class"] == "mlp_chain":
                print("     %-62s n=%d %.1f us  %.0f TF exec" % (k["kernel"], k["launches"], k["us"], k["tflops"]))
    except Exception as e:
        print(v, "ERR", e)
PY
