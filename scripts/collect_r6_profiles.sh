#!/bin/bash
# copy the round-6 evidence from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
F=gpurun_out/r6final; P=profiles
last() { python3 -c "import sys;ls=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')];print(ls[-1])" "$1"; }
[ -f $F/bench_final.json ] && last $F/bench_final.json > $P/r06_bench_final.json
last $F/bench.json > $P/r06_bench_final_run1.json
last $F/bench_again.json > $P/r06_bench_final_run2.json
last $F/bench_fwd_bf16x3.json > $P/r06_bench_ab_bf16x3_everywhere.json
last $F/bench_exstored.json > $P/r06_bench_ab_ex_stored.json
last $F/bench_tn_bf16x3.json > $P/r06_bench_ab_tn_bf16x3.json
last $F/bench_bwd_bf16x3.json > $P/r06_bench_ab_bwd_bf16x3.json
last $F/bench_eager.json > $P/r06_bench_final_eager.json
last $F/bench_256.json > $P/r06_bench_256rays.json
last $F/bench_256_eager.json > $P/r06_bench_256rays_eager.json
last $F/bench_fp32_exact.json > $P/r06_bench_fp32_exact.json
last $F/bench_mixed16.json > $P/r06_bench_mixed16.json
last $F/bench_shipped.json > $P/r06_bench_dtu_shipped.json
last $F/bench_blend.json > $P/r06_bench_garment_blend.json
last $F/bench_cfg5_bf16x3.json > $P/r06_bench_cfg5_1024x256_bf16x3.json
last $F/bench_cfg5_mixed16.json > $P/r06_bench_cfg5_1024x256_mixed16.json
last $F/bench_cfg5_mixed16_exstored.json > $P/r06_bench_cfg5_1024x256_mixed16_ex_stored.json
last $F/bench_strong4096.json > $P/r06_bench_cfg4_4096x128_strong_1gpu.json
cp $F/bench_kernel_stats.csv $P/r06_bench_kernel_stats.csv
cp $F/pmc_mfma_busy.txt $P/r06_pmc_mlp_chain.txt
cp $F/traffic_mlp_chain.json $P/r06_traffic_mlp_chain_bf16x3.json
cp $F/traffic_mlp_chain_cfg5_mixed16.json $P/r06_traffic_mlp_chain_cfg5_mixed16.json
cp $F/traffic_mlp_chain_garment.json $P/r06_traffic_mlp_chain_garment_bf16x3.json
cp $F/traffic_gemm_tn.json $P/r06_traffic_gemm_tn2_f16x2.json
cp $F/step_sequence_graph.txt $P/r06_step_sequence_graph.txt
cp $F/step_sequence_garment_blend.txt $P/r06_step_sequence_garment_blend.txt
cp $F/provenance.txt $P/r06_provenance.txt
cp $F/host_ab.txt $P/r06_host_enqueue_ab.txt
cp $F/chain_timeline.txt $P/r06_chain_timeline.txt
grep -E "passed|failed" $F/pytest_gpu.log | tail -1
ls -la $P/r06_* | wc -l
