#!/bin/bash
# A/B of two builds of libnudf.so on one box: interleaved bench lines (ms_per_step + per-kernel averages) for
# neuraludf_amd/libnudf.so and neuraludf_amd/build/libnudf_$1.so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ab; mkdir -p $O; cd $R
tag=$1
for rep in 1 2 3; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only > $O/new_$rep.json 2> $O/new_$rep.err
  NUDF_LIB=$R/neuraludf_amd/build/libnudf_$tag.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        pk=d.get("roofline",{}).get("per_kernel",{})
        print(f.split("/")[-1], d["ms_per_step"], {k:(v.get("avg_us") if isinstance(v,dict) else v) for k,v in pk.items()})
    except Exception as e: print(f, "ERR", e)
PY
python scripts/chain_hash.py > $O/hash_new.txt 2>&1
NUDF_LIB=$R/neuraludf_amd/build/libnudf_$tag.so python scripts/chain_hash.py > $O/hash_$tag.txt 2>&1
cmp $O/hash_new.txt $O/hash_$tag.txt && echo "chain outputs bit-identical in both builds ($(wc -l < $O/hash_new.txt) tensors)"
python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_chain_rows.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
