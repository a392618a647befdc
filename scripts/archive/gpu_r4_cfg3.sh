#!/bin/bash
# BASELINE config 3 (garment, 1024 rays x 128, pixel + patch blending): bench line + rocprofv3 kernel stats + the ordered
# kernel sequence of one replayed step (attribute the step's time kernel by kernel)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
python bench.py --workload garment_blend_1024x128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_garment.json 2> $O/bench_garment.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python $R/bench.py --workload garment_blend_1024x128 --steps 6 --warmup 3 --no-cpu-baseline --no-forward-only --no-roofline > /dev/null 2>&1
cp $(find /tmp/prof_g -name "*kernel_stats.csv" | head -1) $O/garment_kernel_stats.csv
BENCH_ARGS="--workload garment_blend_1024x128" bash $R/scripts/trace_step_seq.sh > $O/garment_step_seq.txt 2>&1
tail -2 $O/garment_step_seq.txt
