#!/bin/bash
# round 5, call B: the 16-bit-tile chain kernel (bit-identity with the fp32-tile kernel, the fp16 head), the re-run of call A's
# failed tests, and config 5's bench line with the kernel on / off.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5b; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_chain_t16.py tests/test_gpu_mixed16.py tests/test_gpu_status_flag.py tests/test_gpu_jitter_parity.py tests/test_gpu_runner.py "tests/test_gpu_fullsize_parity.py::test_mixed16_vs_oracle_psnr_hierarchical" "tests/test_gpu_fullsize_parity.py::test_mixed16_at_cfg5_shape_vs_reference" -q -s --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest exit $?" >> $O/pytest_new.log
B="--workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --windows 3 --no-cpu-baseline --no-forward-only"
timeout 300 python bench.py $B > $O/cfg5_t16.json 2>> $O/bench.err
NUDF_CHAIN_T16=0 timeout 300 python bench.py $B > $O/cfg5_t16off_head16.json 2>> $O/bench.err
NUDF_CHAIN_T16=0 NUDF_HEAD16=0 timeout 300 python bench.py $B > $O/cfg5_r4.json 2>> $O/bench.err
NUDF_LIB=$R/neuraludf_amd/build/libnudf_t16w2.so timeout 300 python bench.py $B > $O/cfg5_t16_2wg.json 2>> $O/bench.err
timeout 300 python bench.py --precision mixed16 --no-cpu-baseline --windows 3 > $O/headline_mixed16.json 2>> $O/bench.err
python - "$O" <<'PY'
import json, glob, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        pk = {e["kernel"].replace("mlp_chain_kernel", "mck").replace("gemm_tn_group_kernel","tn"): (round(e["us"], 1), round(e.get("frac_mfma", 0), 3)) for e in d["roofline"]["per_kernel"] if e["us"] > 60}
        print(f.split("/")[-1], round(d["ms_per_step"], 3), [round(w, 3) for w in d.get("window_ms", [])], pk)
    except Exception as e:
        print(f, "ERR", e)
PY
NUDF_PRECISION=mixed16 timeout 300 python scripts/chain_timeline.py 65536 > $O/timeline_t16.txt 2>&1
NUDF_PRECISION=mixed16 NUDF_CHAIN_T16=0 NUDF_HEAD16=0 timeout 300 python scripts/chain_timeline.py 65536 > $O/timeline_r4.txt 2>&1
grep -E "passed|failed|error" $O/pytest_new.log | tail -3
