#!/bin/bash
# round 3: resident-workgroup target of the weight-gradient GEMM (one wave of 512 workgroups vs 768 / 1024 / 1536 shorter ones)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3p; mkdir -p $O
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$name.json 2>> $O/bench.err; }
for rep in a b; do
b 512_$rep NUDF_TNG_BLOCKS=512
b 768_$rep NUDF_TNG_BLOCKS=768
b 1024_$rep NUDF_TNG_BLOCKS=1024
b 1536_$rep NUDF_TNG_BLOCKS=1536
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3p"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    tn=[k for k in d["roofline"]["per_kernel"] if "gemm_tn" in k["kernel"]]
    print("%-22s %.3f ms tn %.1f TF %.3f ms | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["gemm_tn"]["tflops"], d["kernels"]["gemm_tn"]["ms"]) + "  ".join("%.0f" % k["us"] for k in tn))
PY
