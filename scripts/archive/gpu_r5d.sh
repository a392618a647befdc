#!/bin/bash
# round 5, call D: wide weight-gradient kernel (bit-identity, float64 accuracy, headline A/B), 16-bit-tile kernel after the
# packed-conversion fix (bit-identity), status / jitter tests, config 5 line.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5d; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_chain_t16.py tests/test_gpu_status_flag.py tests/test_gpu_jitter_parity.py tests/test_gpu_mixed16.py -q -s --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest exit $?" >> $O/pytest_new.log
B="--steps 30 --warmup 8 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg"
for rep in 1 2; do
  timeout 300 python bench.py $B > $O/head_wide_$rep.json 2>> $O/bench.err
  NUDF_TN_FLAGS=1024 timeout 300 python bench.py $B > $O/head_narrow_$rep.json 2>> $O/bench.err
done
C="--workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --windows 3 --no-cpu-baseline --no-forward-only"
timeout 300 python bench.py $C > $O/cfg5_t16.json 2>> $O/bench.err
NUDF_CHAIN_T16=0 NUDF_HEAD16=0 timeout 300 python bench.py $C > $O/cfg5_r4.json 2>> $O/bench.err
python - "$O" <<'PY'
import json, glob, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        pk = {e["kernel"].replace("mlp_chain_kernel", "mck").replace("gemm_tn_group_kernel","tn"): (round(e["us"], 1), round(e.get("frac_mfma", 0), 3)) for e in d["roofline"]["per_kernel"] if e["us"] > 100}
        print(f.split("/")[-1], round(d["ms_per_step"], 3), [round(w, 3) for w in d.get("window_ms", [])], pk)
    except Exception as e:
        print(f, "ERR", e)
PY
grep -E "passed|failed|error" $O/pytest_new.log | tail -3
