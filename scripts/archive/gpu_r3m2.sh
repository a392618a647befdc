#!/bin/bash
# full GPU suite + smoke + the default bench line (+ rocprofv3 stats of the same command) of the current tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m2
mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -n 3 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -n 1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-forward-only > $O/prof_bench.log 2>&1)
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3m2"
d=json.load(open(O+"/bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["cpu_baseline"], d.get("psnr_vs_ref",{}).get("value_db"))
for k in d["roofline"]["per_kernel"]: print("  %-70s %5.0f us %.3f" % (k["kernel"][:70], k["us"], k["frac_mfma"]))
PY
