#!/bin/bash
# PMC passes (one counter set per pass, kernel trace only) over scripts/tn3_traffic.py; per-dispatch means per launch group
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_tn3
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o p -- python $R/scripts/tn3_traffic.py > $OUT/$tag.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
OUT = sys.argv[1]
names = ["one_tile", "two_tiles", "four_tiles", "udf_adjoint"]
for f in sorted(glob.glob(OUT + "/*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "gemm_tn3_group_kernel" in r["Kernel_Name"]]
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in per.items():
        v.sort()
        vals = [x[1] for x in v]
        groups = [vals[4 * i:4 * i + 4] for i in range(len(vals) // 4)]
        print(c, {names[i] if i < len(names) else i: round(sum(g) / len(g), 1) for i, g in enumerate(groups)})
PY
grep -h "operand bytes" $OUT/FETCH_SIZE.log
