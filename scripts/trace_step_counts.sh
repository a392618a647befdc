#!/bin/bash
# kernel launches of ONE steady-state train step of the headline workload (between two consecutive adam kernels), by name
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof7
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof7 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-forward-only --no-roofline > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/prof7/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
idx = [i for i, k in enumerate(ks) if k[0].startswith("adam")]
a, b = idx[-2], idx[-1]
c = collections.Counter(); t = collections.Counter()
for k, us in ks[a + 1:b + 1]:
    n = k.split("(")[0][:70]
    if "FillFunctor" in k: n = "torch fill"
    elif "at::native" in k: n = "torch " + k.split("at::native::")[1][:48]
    c[n] += 1; t[n] += us
print("launches in the step:", b - a, " kernel time %.1f us" % sum(t.values()))
for n, v in sorted(c.items(), key=lambda kv: -t[kv[0]]):
    print("%4d x %-72s %9.1f us" % (v, n, t[n]))
PY
