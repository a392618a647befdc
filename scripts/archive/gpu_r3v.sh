#!/bin/bash
# A/B, fp32 headline workload: start offset of the odd wave slots in the shared-tile kernel (shipping: 2 sleeps) vs none.
# Build the variant first: scripts/build_variants.sh stag32_0 mlp_chain.hip -DNUDF_STAGGER32=0u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3v
mkdir -p $O
B=$GRAFT_REPO_ROOT/neuraludf_amd/build
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
for r in a b c; do
b stag2_$r NUDF_X=1
b stag0_$r NUDF_LIB=$B/libnudf_stag32_0.so
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3v"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-22s %.3f ms chain %.2f ms tn %.2f ms | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]) + "  ".join("%s %.0f" % (k["kernel"].split()[2][:8], k["us"]) for k in d["roofline"]["per_kernel"] if "chain_kernel" in k["kernel"]))
PY
