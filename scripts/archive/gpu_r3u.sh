#!/bin/bash
# A/B at the config-5 shape (16-bit mode): start offset of the odd wave slots, in s_sleep(127) units (shipping: 0).
# Build the variants first: for k in 1 2 3 4; do scripts/build_variants.sh stag$k mlp_chain.hip -DNUDF_STAGGER16=${k}u; done
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3u
mkdir -p $O
B=$GRAFT_REPO_ROOT/neuraludf_amd/build
b() { name=$1; shift; env "$@" timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
for r in a b; do
b stag0_$r NUDF_X=1
for k in 1 2 3 4; do b stag${k}_$r NUDF_LIB=$B/libnudf_stag$k.so; done
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3u"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-22s %.3f ms chain %.2f ms tn %.2f ms | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]) + "  ".join("%s %.0f" % (k["kernel"].split()[2][:8], k["us"]) for k in d["roofline"]["per_kernel"] if "chain" in k["kernel"]))
PY
