#!/bin/bash
# colour-net chains on 32-point tiles (NUDF_COLOR_TILE=32) against the automatic choice: fp32 headline and the 16-bit config-5 shape
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $O; rm -f $O/*
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
c() { name=$1; shift; env "$@" timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
for r in a b; do
b f32_auto_$r NUDF_X=1
b f32_c32_$r NUDF_COLOR_TILE=32
c m16_auto_$r NUDF_X=1
c m16_c32_$r NUDF_COLOR_TILE=32
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3w"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-22s %.3f ms chain %.2f ms | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"]) + "  ".join("%s %.0f" % (k["kernel"].split()[2][:8], k["us"]) for k in d["roofline"]["per_kernel"] if "relu" in k["kernel"]))
PY
