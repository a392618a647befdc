#!/bin/bash
# A/B of two builds on one box, per-launch-class times (bench.py per_kernel): $1 = variant tag
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ab2; mkdir -p $O; cd $R
tag=$1
for rep in 1 2; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-forward-only > $O/base_$rep.json 2> $O/base_$rep.err
  NUDF_LIB=$R/neuraludf_amd/build/libnudf_$tag.so python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-forward-only > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"],3))
    for k in d["roofline"]["per_kernel"]:
        if "chain" in k["kernel"]: print("     %-70s n=%d  %.1f us each" % (k["kernel"][:70], k["launches"], k["us"]/k["launches"]))
PY
