#!/bin/bash
# sanity run of bench.py after the config.operand_mode addition
mkdir -p gpurun_out/r6af
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > gpurun_out/r6af/bench.json 2> gpurun_out/r6af/bench.err; echo "exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6af/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["operand_mode"], d["roofline"]["traffic_stale"])
PY
