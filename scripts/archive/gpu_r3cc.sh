#!/bin/bash
# transposed-product K loops: next group's operand requests spread over the MFMA quarters (NUDF_TQ_SPREAD=1) against all in front (=0)
# build the variant first: scripts/build_variants.sh nospread mlp_chain_rows.hip -DNUDF_TQ_SPREAD=0
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3cc
mkdir -p $O; rm -f $O/*
B=$GRAFT_REPO_ROOT/neuraludf_amd/build
timeout 600 python -m pytest tests/test_gpu_chain_rows.py -q -x > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -n 2 $O/pytest.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
for r in a b c; do
b spread_$r NUDF_X=1
b nospread_$r NUDF_LIB=$B/libnudf_nospread.so
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3cc"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-18s %.3f ms chain %.2f ms frac %.3f | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"], d["roofline"]["frac"]) + "  ".join("%s %.0f" % (k["kernel"].split()[1][:3], k["us"]) for k in d["roofline"]["per_kernel"] if "tq_kernel" in k["kernel"]))
PY
