#!/bin/bash
# one gpurun call: A/B of the two fused-chain kernels (correctness, per-launch times, timelines)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
timeout 600 python scripts/chain_ab.py check time timeline > gpurun_out/ab/chain_ab.log 2>&1
echo "exit $?" >> gpurun_out/ab/chain_ab.log
AB_TILES=128 NUDF_LIB=$GRAFT_REPO_ROOT/neuraludf_amd/libnudf_nosync.so timeout 300 python scripts/chain_ab.py time > gpurun_out/ab/chain_ab_nosync.log 2>&1
echo "exit $?" >> gpurun_out/ab/chain_ab_nosync.log
AB_TILES=128 NUDF_CHAIN_WIN2=3 timeout 300 python scripts/chain_ab.py time > gpurun_out/ab/chain_ab_win3.log 2>&1
NUDF_CHAIN_ROWS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/bench_rows0.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/bench_rows1.log 2>&1
for f in gpurun_out/ab/*.log; do echo "== $f"; grep -v Warn $f | tail -n 4 | cut -c1-300; done
