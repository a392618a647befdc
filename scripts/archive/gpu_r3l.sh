#!/bin/bash
# A/B at the config-5 shape (1024 x 256, 16-bit mode) between the shipping library and variant builds of it:
#   scripts/build_variants.sh <tag> <source.hip> -D<SWITCH>=<value>      (e.g. noring mlp_chain.hip -DNUDF_MMA16_RING=0)
#   echo "<tag> ..." > neuraludf_amd/build/ab_variants.txt; scripts/gpurun.sh scripts/gpu_r3l.sh     (the box sees files, not the caller's environment)
# (used for profiles/r03_chain_experiments.txt items 7-9: NUDF_SEQ16, NUDF_MMA16_RING, the removed packed-pair algebra)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $O
B=$GRAFT_REPO_ROOT/neuraludf_amd/build
VARIANTS=${VARIANTS:-$(cat $B/ab_variants.txt 2>/dev/null)}
timeout 600 python -m pytest tests/test_gpu_mixed16.py "tests/test_gpu_fullsize_parity.py::test_mixed16_at_cfg5_shape_vs_reference" "tests/test_gpu_fullsize_parity.py::test_mixed16_vs_oracle_psnr_hierarchical" -q -s > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed" $O/pytest.log | tail -n 2
b() { name=$1; shift; env "$@" timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$name.json 2>> $O/bench.err; }
for r in a b; do
  b ship_$r NUDF_X=1
  for v in $VARIANTS; do b ${v}_$r NUDF_LIB=$B/libnudf_$v.so; done
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3l"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-22s %.3f ms chain %.2f ms tn %.2f ms | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]) + "  ".join("%s %.0f" % (k["kernel"].split()[2][:8], k["us"]) for k in d["roofline"]["per_kernel"] if "chain" in k["kernel"]))
PY
