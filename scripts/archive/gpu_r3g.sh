#!/bin/bash
# NOTE: records an experiment of round 3 whose code was removed again (NUDF_CHAIN_RING / NUDF_CHAIN_W8 / NUDF_COLOR_TILE / NUDF_SEQ16
# switches no longer exist); kept as the provenance of profiles/r03_chain_experiments.txt.
# round 3, call G: 8 waves per workgroup for 32-point tiles / narrow chains (NUDF_CHAIN_W8 = 0 / 1 / 2 / 3), tq ring default,
# chain-kernel parity suites incl. the paired kernel's bit-identity
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain_rows.py tests/test_gpu_kernels.py tests/test_gpu_edges.py -q > $O/pytest_w8.log 2>&1
echo "pytest rc $?" >> $O/pytest_w8.log
tail -n 5 $O/pytest_w8.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$name.json 2>> $O/bench.err; }
b w8_0 NUDF_CHAIN_W8=0
b w8_1 NUDF_CHAIN_W8=1
b w8_2 NUDF_CHAIN_W8=2
b w8_3 NUDF_CHAIN_W8=3
b w8_0b NUDF_CHAIN_W8=0
b w8_3b NUDF_CHAIN_W8=3
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3g"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print(os.path.basename(f), "%.3f ms" % d["ms_per_step"], "chain %.1f TF" % d["kernels"]["mlp_chain"]["tflops"], "tn %.1f TF" % d["kernels"]["gemm_tn"]["tflops"], "fwd-only %.3f" % d["forward_only"]["ms"])
    for k in d["roofline"]["per_kernel"]:
        if "chain" in k["kernel"]: print("    %-70s %7.1f us %6.1f TF" % (k["kernel"], k["us"], k["tflops"]))
PY
