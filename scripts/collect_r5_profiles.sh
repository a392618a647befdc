#!/bin/bash
# copy the round-5 evidence from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
F=gpurun_out/r5final; P=profiles
last() { python3 -c "import sys;ls=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')];print(ls[-1])" "$1"; }
last $F/bench.json > $P/r05_bench_final.json
last $F/bench_eager.json > $P/r05_bench_final_eager.json
last $F/bench_256.json > $P/r05_bench_256rays.json
last $F/bench_fp32_exact.json > $P/r05_bench_fp32_exact.json
last $F/bench_mixed16.json > $P/r05_bench_mixed16.json
last $F/bench_shipped.json > $P/r05_bench_dtu_shipped.json
last $F/bench_blend.json > $P/r05_bench_garment_blend.json
last $F/bench_cfg5_bf16x3.json > $P/r05_bench_cfg5_1024x256_bf16x3.json
last $F/bench_cfg5_mixed16.json > $P/r05_bench_cfg5_1024x256_mixed16.json
last $F/bench_cfg5_mixed16_r4kernels.json > $P/r05_bench_cfg5_1024x256_mixed16_r4kernels.json
last $F/bench_strong4096.json > $P/r05_bench_cfg4_4096x128_strong_1gpu.json
cp $F/bench_kernel_stats.csv $P/r05_bench_kernel_stats.csv
cp $F/pmc_mfma_busy.txt $P/r05_pmc_mlp_chain.txt
cp $F/traffic_mlp_chain.json $P/r05_traffic_mlp_chain_bf16x3.json
cp $F/traffic_mlp_chain_cfg5_mixed16.json $P/r05_traffic_mlp_chain_cfg5_mixed16.json
cp $F/traffic_mlp_chain_garment.json $P/r05_traffic_mlp_chain_garment_bf16x3.json
cp $F/step_sequence_graph.txt $P/r05_step_sequence_graph.txt
cp $F/step_sequence_garment_blend.txt $P/r05_step_sequence_garment_blend.txt
cp $F/provenance.txt $P/r05_provenance.txt
[ -f gpurun_out/r5b/timeline_t16.txt ] && grep -v Warning gpurun_out/r5b/timeline_t16.txt > $P/r05_chain_timeline_mixed16.txt
[ -f gpurun_out/r5f/tnw_phases.txt ] && grep -v "Warning\|amdgpu.ids" gpurun_out/r5f/tnw_phases.txt > $P/r05_tn_wide_phases.txt
grep -E "passed|failed" $F/pytest_gpu.log | tail -1
ls -la $P/r05_* | wc -l
