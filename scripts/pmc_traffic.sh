#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) for kernels matching $1 of the command after "--"; prints per-dispatch KB
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift; shift
OUT=$R/gpurun_out/traffic_$K
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$set -o p -- "$@" > $OUT/$set.log 2>&1
done
python - "$K" "$OUT" <<'PY'
import csv, glob, sys, collections, json
K, OUT = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(OUT + "/*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if K in row["Kernel_Name"]:
            per[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in per.items():
        res.setdefault(k[0], {})[k[1]] = {"n": len(v), "mean_KB": sum(v) / len(v), "list_KB": [round(x) for x in v[:40]]}
print(json.dumps(res, indent=1))
PY
