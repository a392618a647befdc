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
python - "$K" "$OUT" "$R" <<'PY'
import csv, glob, sys, collections, json, subprocess, hashlib
K, OUT, R = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, R)
from neuraludf_amd import build as _b
# what the counters were recorded ON: bench.py compares the digest with the tree it runs from (`roofline.traffic_stale`)
res = {"_recorded_on": {"kernel": K, "source_digest": _b.source_digest(K), "sources": _b.KERNEL_SOURCES.get(K),
                        "so_sha256": hashlib.sha256(open(_b.LIB, "rb").read()).hexdigest()}}
for f in sorted(glob.glob(OUT + "/*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if K in row["Kernel_Name"]:
            per[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in per.items():
        res.setdefault(k[0], {})[k[1]] = {"n": len(v), "mean_KB": sum(v) / len(v), "list_KB": [round(x) for x in v[:40]]}
print(json.dumps(res, indent=1))
PY
