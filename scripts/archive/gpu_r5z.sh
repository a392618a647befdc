#!/bin/bash
# extra round-5 evidence on the frozen build: HBM traffic of the weight-gradient kernel inside the real step (PMC), rocprofv3 kernel
# stats of config 3 and config 5 (mixed16)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5z; rm -rf $O; mkdir -p $O
cd $R
bash scripts/pmc_traffic.sh gemm_tn3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-forward-only --no-fp32-leg --no-power --graph 0 > $O/traffic_gemm_tn3.json 2> $O/traffic.err
for w in "garment_blend_1024x128:bf16x3:garment_blend" "dtu_scan24_1024x256:mixed16:cfg5_mixed16"; do
  IFS=: read wl prec tag <<< "$w"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o bench -- python $R/bench.py --workload $wl --precision $prec --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-forward-only --no-fp32-leg --no-power > $O/prof_$tag.log 2>&1)
  find $O/prof_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
  rm -rf $O/prof_$tag
done
cat $O/traffic_gemm_tn3.json | head -30
head -8 $O/kernel_stats_garment_blend.csv; head -8 $O/kernel_stats_cfg5_mixed16.csv
