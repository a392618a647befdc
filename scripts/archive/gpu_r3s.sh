#!/bin/bash
# 4-point packed 16-bit state: parity tests, then the config-5 bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mixed16.py tests/test_gpu_kernels.py tests/test_abi.py -k "mixed16 or gemm_tn or color or abi" "tests/test_gpu_fullsize_parity.py::test_mixed16_at_cfg5_shape_vs_reference" "tests/test_gpu_fullsize_parity.py::test_mixed16_vs_oracle_psnr_hierarchical" -q -x > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -n 30 $O/pytest.log
for i in a b; do
timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_p4_$i.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3s"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-22s %.3f ms chain %.2f ms tn %.2f ms | " % (os.path.basename(f), d["ms_per_step"], d["kernels"]["mlp_chain"]["ms"], d["kernels"]["gemm_tn"]["ms"]) + "  ".join("%s %.0f" % (k["kernel"].split()[2][:8], k["us"]) for k in d["roofline"]["per_kernel"]))
PY
