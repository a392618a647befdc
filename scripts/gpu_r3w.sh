#!/bin/bash
# 16-bit mode at the config-5 shape: 64-point tiles (2 workgroups / CU) against 32-point tiles (3 / CU by registers)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $O
b() { name=$1; shift; env "$@" timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
for r in a b; do
b t64_$r NUDF_X=1
b t32_$r NUDF_CHAIN_TILE=32
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3w"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-22s %.3f ms chain %.2f ms tn %.2f ms | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]) + "  ".join("%s %.0f" % (k["kernel"].split()[2][:8], k["us"]) for k in d["roofline"]["per_kernel"] if "chain" in k["kernel"]))
PY
