#!/bin/bash
# in-box A/B of two TREES (this one and the worktree in $1): interleaved default bench lines, graph and eager
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4abt; mkdir -p $O
for rep in 1 2 3; do
  for t in new old; do
    if [ $t = new ]; then d=$R; else d=$R/$1; fi
    (cd $d && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only --no-roofline > $O/${t}_$rep.json 2> $O/${t}_$rep.err)
    (cd $d && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only --no-roofline --graph 0 > $O/${t}_eager_$rep.json 2>> $O/${t}_$rep.err)
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"],4))
    except Exception as e: print(f, "ERR", e)
PY
