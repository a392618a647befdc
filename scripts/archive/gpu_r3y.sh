#!/bin/bash
# transposed-product kernel: X1 streamed one tile ahead (NUDF_TQ_S1) -- bit-identity tests, then the step with 0 / 1 / 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3y
mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_chain_rows.py -q -x > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -n 3 $O/pytest.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
for r in a b; do
b s0_$r NUDF_TQ_S1=0
b s1_$r NUDF_TQ_S1=1
b s2_$r NUDF_TQ_S1=2
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3y"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-16s %.3f ms chain %.2f ms frac %.3f | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"], d["roofline"]["frac"]) + "  ".join("%s %.0f" % (k["kernel"].split()[1][:3], k["us"]) for k in d["roofline"]["per_kernel"] if "tq_kernel" in k["kernel"]))
PY
