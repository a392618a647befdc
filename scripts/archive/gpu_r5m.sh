#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5final; mkdir -p $O; cd $R
for i in 1 2; do timeout 600 python bench.py > $O/bench_rerun_$i.json 2>> $O/bench.err; done
timeout 300 python bench.py --workload garment_blend_1024x128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_blend.json 2>> $O/bench.err
python - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5final"
for f in ("bench_rerun_1","bench_rerun_2","bench_blend"):
    d=json.loads(open(O+"/"+f+".json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],3), [round(w,3) for w in d["window_ms"]], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.get("fp32_exact",{}).items() if k in("ms_per_step","frac","window_ms","launch","error")})
PY
