#!/bin/bash
# round 3, call H: wave-priority / start-up delay variants of the transposed-product kernel (A/B libraries)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $O
B=$GRAFT_REPO_ROOT/neuraludf_amd/build
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$name.json 2>> $O/bench.err; }
for rep in a b; do
b base_$rep NUDF_X=1
b noprio_$rep NUDF_LIB=$B/libnudf_noprio.so
b nosleep_$rep NUDF_LIB=$B/libnudf_nosleep.so
b noprio_nosleep_$rep NUDF_LIB=$B/libnudf_noprio_nosleep.so
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3h"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    tq=[k for k in d["roofline"]["per_kernel"] if "tq_kernel" in k["kernel"]]
    print("%-28s %.3f ms chain %.1f TF tn %.1f TF | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["tflops"], d["kernels"]["gemm_tn"]["tflops"]) + "  ".join("%s %.0f" % (k["kernel"].split()[1][:8], k["us"]) for k in tq))
PY
