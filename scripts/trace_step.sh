cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof6 -o b -- python $GRAFT_REPO_ROOT/bench.py ${TRACE_ARGS:---workload dtu_scan24_1024x256 --precision mixed16} --steps 2 --warmup 1 --no-cpu-baseline --no-forward-only --no-roofline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof6/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:34], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
# last step: find the last adam kernel and print the kernels between the previous adam and it
idx = [i for i, k in enumerate(ks) if k[0].startswith("adam")]
a, b = idx[-2], idx[-1]
tot = 0
for k, us in ks[a + 1:b + 1]:
    if us > 30:
        print(f"{k:36s} {us:9.1f} us")
    tot += us
print("sum of kernel time in the step: %.1f us" % tot)
PY
