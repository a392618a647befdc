cd $GRAFT_REPO_ROOT
for v in "" "NUDF_BLOCKED_STATE=0" "NUDF_TN_FLAGS=8" "NUDF_BLOCKED_STATE=0 NUDF_TN_FLAGS=8" "NUDF_TN_FLAGS=16"; do
  echo "== $v"
  env $v python bench.py --rays-per-gpu 256 --no-cpu-baseline 2>/dev/null | python scripts/bench_kernels_line.py
done
