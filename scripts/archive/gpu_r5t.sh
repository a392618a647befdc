#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5t; rm -rf $O; mkdir -p $O
cd $R
python - > $O/dev.txt 2>&1 <<'PY'
import torch, os, glob
pr = torch.cuda.get_device_properties(0)
print("pci %04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
    print(h, os.path.realpath(os.path.dirname(os.path.dirname(h))), [ (n, open(os.path.join(h,n)).read().strip()) for n in ("power1_average","power1_input","freq1_input") if os.path.exists(os.path.join(h,n))])
PY
cat $O/dev.txt | tail -10
timeout 300 python bench.py --steps 30 --warmup 8 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg > $O/head.json 2>> $O/bench.err
timeout 300 python bench.py --precision mixed16 --workload dtu_scan24_1024x256 --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg > $O/cfg5.json 2>> $O/bench.err
timeout 300 python bench.py --precision fp32 --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-forward-only --no-fp32-leg > $O/fp32.json 2>> $O/bench.err
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get("power") or {}
        print(f.split("/")[-1], round(d["ms_per_step"], 3), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in p.items() if k != "what"})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/bench.err
