#!/bin/bash
# NOTE: records an experiment of round 3 whose code was removed again (NUDF_CHAIN_RING / NUDF_CHAIN_W8 / NUDF_COLOR_TILE / NUDF_SEQ16
# switches no longer exist); kept as the provenance of profiles/r03_chain_experiments.txt.
# round 3, call D: K-loop pipeline depth (NUDF_CHAIN_RING 0 / 3 / 4), ring for the tq sweeps (NUDF_TQ_RING=3), colour-net
# tile size; per-kernel lines of the bench for each; bit-identity of the ring kernels vs the two-set loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$name.json 2>> $O/bench.err; }
b ring0 NUDF_CHAIN_RING=0
b ring4 NUDF_CHAIN_RING=4
b ring3 NUDF_CHAIN_RING=3
b ring4_tq3 NUDF_CHAIN_RING=4 NUDF_TQ_RING=3
b ring4_col32 NUDF_CHAIN_RING=4 NUDF_COLOR_TILE=32
b ring0_b NUDF_CHAIN_RING=0
NUDF_CHAIN_RING=4 NUDF_TQ_RING=3 timeout 600 python -m pytest tests/test_gpu_chain_rows.py tests/test_gpu_kernels.py tests/test_gpu_edges.py -q -x > $O/pytest_ring.log 2>&1
echo "pytest rc $?" >> $O/pytest_ring.log
tail -n 3 $O/pytest_ring.log
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3d"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print(os.path.basename(f), "%.3f ms" % d["ms_per_step"], "chain %.1f TF" % d["kernels"]["mlp_chain"]["tflops"], "tn %.1f TF" % d["kernels"]["gemm_tn"]["tflops"])
    for k in d["roofline"]["per_kernel"]:
        print("    %-70s %7.1f us %6.1f TF" % (k["kernel"], k["us"], k["tflops"]))
PY
