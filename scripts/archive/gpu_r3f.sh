#!/bin/bash
# round 3, call F: paired-tile chain kernel -- bit-identity tests, bench with NUDF_CHAIN_PAIR = 0 / 1 / 2, timelines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain_rows.py -q -x -k "pair" > $O/pytest_pair.log 2>&1
echo "pytest rc $?" >> $O/pytest_pair.log
tail -n 4 $O/pytest_pair.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$name.json 2>> $O/bench.err; }
b pair0 NUDF_CHAIN_PAIR=0
b pair1 NUDF_CHAIN_PAIR=1
b pair2 NUDF_CHAIN_PAIR=2
b pair0_b NUDF_CHAIN_PAIR=0
b pair2_b NUDF_CHAIN_PAIR=2
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3f"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print(os.path.basename(f), "%.3f ms" % d["ms_per_step"], "chain %.1f TF" % d["kernels"]["mlp_chain"]["tflops"], "tn %.1f TF" % d["kernels"]["gemm_tn"]["tflops"], "fwd-only %.3f" % d["forward_only"]["ms"])
    for k in d["roofline"]["per_kernel"]:
        print("    %-70s %7.1f us %6.1f TF" % (k["kernel"], k["us"], k["tflops"]))
PY
TIMELINE_COLOUR=1 timeout 300 python scripts/chain_timeline.py 65536 > $O/timeline_colour_pair.txt 2>&1
timeout 300 python scripts/chain_timeline.py 65536 > $O/timeline_udf_pair.txt 2>&1
grep -v Warning $O/timeline_colour_pair.txt | grep -E "launch|per wave|MFMA ticks" | head -n 20
grep -v Warning $O/timeline_udf_pair.txt | grep -E "launch|per wave|MFMA ticks" | head -n 20
