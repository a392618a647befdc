#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5i; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_blend_loss.py tests/test_gpu_blend.py tests/test_gpu_graph.py tests/test_gpu_step_loss.py tests/test_gpu_runner.py "tests/test_gpu_fullsize_parity.py::test_cfg3_mix_sampling_and_blending_vs_reference" "tests/test_gpu_fullsize_parity.py::test_cfg3_garment_geometry_1024_rays_vs_reference" -q -s --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest exit $?" >> $O/pytest_new.log
B="--workload garment_blend_1024x128 --steps 10 --warmup 3 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg --no-roofline"
for rep in 1 2; do
  timeout 300 python bench.py $B > $O/blend_fused_$rep.json 2>> $O/bench.err
  NUDF_FUSE_BLEND_LOSS=0 timeout 300 python bench.py $B > $O/blend_generic_$rep.json 2>> $O/bench.err
done
BENCH_ARGS="--workload garment_blend_1024x128" bash scripts/trace_step_seq.sh > $O/step_sequence_garment_blend.txt 2>&1
python - "$O" <<'PY'
import json, glob, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), [round(w, 3) for w in d.get("window_ms", [])])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -1 $O/step_sequence_garment_blend.txt
grep -E "fused blend|device launches|passed|failed|error" $O/pytest_new.log | tail -8
