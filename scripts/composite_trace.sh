#!/bin/bash
# kernel-only durations of the composite kernels at the roofline sizes (rocprofv3 kernel trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ctrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for n in 8192 32768; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$n -o t -- python $R/scripts/composite_prof.py $n > $OUT/n$n.log 2>&1
  f=$(find $OUT/n$n -name "*kernel_stats.csv" | head -1)
  echo "N=$n"; grep -E "composite|partial_sums" $f | cut -d, -f1-4,6,7 | cut -c1-160
  find $OUT/n$n -name "*kernel_trace.csv" -delete
done
