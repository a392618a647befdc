#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5h; rm -rf $O; mkdir -p $O
cd $R
B="--steps 30 --warmup 8 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg"
for tag in ship cust2 cust4 xcst3 ship; do
  if [ $tag = ship ]; then lib=$R/neuraludf_amd/libnudf.so; else lib=$R/neuraludf_amd/build/libnudf_$tag.so; fi
  NUDF_LIB=$lib timeout 300 python bench.py $B > $O/probe_${tag}_$RANDOM.json 2>> $O/probe.err
done
python - "$O" <<'PY'
import json, glob, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        pk = {e["kernel"].replace("mlp_chain_kernel", "mck"): round(e["us"], 1) for e in d["roofline"]["per_kernel"] if e["class"] == "mlp_chain"}
        print(f.split("/")[-1], round(d["ms_per_step"], 3), [round(w, 3) for w in d.get("window_ms", [])], pk)
    except Exception as e:
        print(f, "ERR", e)
PY
