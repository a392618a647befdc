#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16x3.py "tests/test_gpu_fullsize_parity.py::test_cfg2_render_core_and_all_parameter_gradients_vs_reference" -q --tb=short -p no:cacheprovider -x > $O/pytest_new.log 2>&1; echo "pytest exit $?" >> $O/pytest_new.log
grep -E "passed|failed|error" $O/pytest_new.log | tail -3
NUDF_LIB=$R/neuraludf_amd/build/libnudf_stamps.so timeout 300 python scripts/tn3_phases.py 65536 2>&1 | grep -v Warn | tail -6
B="--steps 30 --warmup 8 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg"
for rep in 1 2; do
  timeout 300 python bench.py $B > $O/head_buf_$rep.json 2>> $O/bench.err
  NUDF_LIB=$R/neuraludf_amd/build/libnudf_flat.so timeout 300 python bench.py $B > $O/head_flat_$rep.json 2>> $O/bench.err
  NUDF_LIB=$R/neuraludf_amd/build/libnudf_oldplan.so timeout 300 python bench.py $B > $O/head_oldplan_$rep.json 2>> $O/bench.err
done
python - "$O" <<'PY'
import json, glob, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        pk = {e["kernel"].replace("gemm_tn_group_kernel","tn"): (round(e["us"], 1), round(e.get("frac_mfma", 0), 3)) for e in d["roofline"]["per_kernel"] if e["class"] == "gemm_tn"}
        print(f.split("/")[-1], round(d["ms_per_step"], 3), [round(w, 3) for w in d.get("window_ms", [])], pk)
    except Exception as e:
        print(f, "ERR", e)
PY
