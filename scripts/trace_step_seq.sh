#!/bin/bash
# ordered kernel sequence (name, duration, gap to the previous kernel) of ONE steady-state step (between two adam kernels);
# BENCH_ARGS selects eager / graph etc.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof8
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof8 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-forward-only --no-roofline --no-fp32-leg ${BENCH_ARGS:-} > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof8/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam")]
a, b = idx[-2], idx[-1]
prev_end = int(rows[a]["End_Timestamp"])
tot = gaps = 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    if "at::native" in n: n = "torch:" + n.split("at::native::")[1][:60]
    print("%8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, (s - prev_end) / 1e3, n[:90]))
    tot += (e - s) / 1e3; gaps += max(0, (s - prev_end) / 1e3); prev_end = e
print("kernels %.1f us, gaps %.1f us, launches %d" % (tot, gaps, b - a))
PY
