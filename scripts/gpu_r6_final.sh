#!/bin/bash
# round 6 end-of-round evidence: full GPU test suite, smoke, the default bench line (+ rocprofv3 kernel stats of the same
# command and flags), eager A/B, secondary workloads, PMC (MFMA-busy of the chain kernels, HBM traffic per chain launch).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6final; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --graph 0 --no-cpu-baseline > $O/bench_eager.json 2>> $O/bench.err
timeout 300 python bench.py --rays-per-gpu 256 --no-cpu-baseline --no-roofline > $O/bench_256.json 2>> $O/bench.err
timeout 300 python bench.py --rays-per-gpu 256 --graph 0 --no-cpu-baseline --no-roofline > $O/bench_256_eager.json 2>> $O/bench.err
timeout 300 python bench.py --precision fp32 --no-cpu-baseline > $O/bench_fp32_exact.json 2>> $O/bench.err
timeout 300 python bench.py --precision mixed16 --no-cpu-baseline > $O/bench_mixed16.json 2>> $O/bench.err
timeout 300 python bench.py --workload "dtu_shipped_512x114+32" --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_shipped.json 2>> $O/bench.err
timeout 300 python bench.py --workload garment_blend_1024x128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_blend.json 2>> $O/bench.err
timeout 300 python bench.py --workload dtu_scan24_1024x256 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_bf16x3.json 2>> $O/bench.err
timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_mixed16.json 2>> $O/bench.err
NUDF_EX_FLY=0 timeout 300 python bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_mixed16_exstored.json 2>> $O/bench.err
# round-6 A/B legs at the headline: six bf16 products on the forward-order sweeps (round 5's arithmetic), the stored second-order
# term, bf16x3 weight-gradient GEMMs, bf16x3 backward sweeps
NUDF_FWD_F16X2=0 timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_fwd_bf16x3.json 2>> $O/bench.err
NUDF_EX_FLY=0 timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_exstored.json 2>> $O/bench.err
NUDF_TN_F16X2=0 timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_tn_bf16x3.json 2>> $O/bench.err
NUDF_BWD_F16X2=0 timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_bwd_bf16x3.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg > $O/bench_again.json 2>> $O/bench.err
timeout 300 python bench.py --workload dtu_scan118_4096x128 --steps 5 --warmup 2 --windows 3 --no-cpu-baseline --no-fp32-leg > $O/bench_strong4096.json 2>> $O/bench.err
# rocprofv3 kernel trace + stats of the default command (same build, same flags)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-forward-only --no-fp32-leg > $O/prof_bench.log 2>&1)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
# PMC: MFMA-busy of the chain kernels over the bench command (separate pass, kernel-trace only)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_busy -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-forward-only --no-fp32-leg --graph 0 > $O/pmc_busy.log 2>&1)
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_busy/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "mlp_chain" in n or "gemm_tn" in n:
            agg[n.split("(")[0][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/pmc_mfma_busy.txt", "w") as out:
    out.write("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (bench.py --steps 4 --warmup 2 --graph 0), per dispatch means;\n"
              "MFMA-busy %% = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)\n")
    for k, c in sorted(agg.items()):
        b, g = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
        if b and g:
            mb, mg = sum(b) / len(b), sum(g) / len(g)
            out.write("%-50s n=%3d  mfma_busy_cycles %.3e  gui_active %.3e  -> MFMA-busy %.1f %%\n" % (k, len(b), mb, mg, mb / 1024 / (mg / 8) * 100))
print(open(O + "/pmc_mfma_busy.txt").read())
PY
rm -rf $O/pmc_busy
# HBM traffic per chain launch (FETCH_SIZE / WRITE_SIZE in separate passes)
bash scripts/pmc_traffic.sh mlp_chain -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-forward-only --no-fp32-leg --graph 0 > $O/traffic_mlp_chain.json 2> $O/traffic.err
bash scripts/pmc_traffic.sh mlp_chain -- python $R/bench.py --workload dtu_scan24_1024x256 --precision mixed16 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-forward-only --graph 0 > $O/traffic_mlp_chain_cfg5_mixed16.json 2>> $O/traffic.err
bash scripts/pmc_traffic.sh mlp_chain -- python $R/bench.py --workload garment_blend_1024x128 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-forward-only --no-fp32-leg --graph 0 > $O/traffic_mlp_chain_garment.json 2>> $O/traffic.err
bash scripts/pmc_traffic.sh gemm_tn -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-forward-only --no-fp32-leg --graph 0 > $O/traffic_gemm_tn.json 2>> $O/traffic.err
for rays in 256 512; do for fast in 0 1; do echo -n "rays $rays NUDF_HOST_FAST=$fast: "; NUDF_HOST_FAST=$fast timeout 300 python scripts/host_profile.py $rays 2>&1 | grep "host enqueue"; done; done > $O/host_ab.txt
timeout 300 python scripts/chain_timeline.py 65536 2>&1 | grep -v "Warn\|amdgpu.ids\|distinct" > $O/chain_timeline.txt
BENCH_ARGS="--workload garment_blend_1024x128" bash scripts/trace_step_seq.sh > $O/step_sequence_garment_blend.txt 2>&1
BENCH_ARGS="" bash scripts/trace_step_seq.sh > $O/step_sequence_graph.txt 2>&1
(nproc; grep -m1 "model name" /proc/cpuinfo; rocm-smi --showproductname 2>/dev/null | head -12; git -C $R rev-parse HEAD 2>/dev/null; sha256sum $R/neuraludf_amd/libnudf.so) > $O/provenance.txt 2>&1
tail -n 3 $O/pytest_gpu.log; tail -n 1 $O/smoke.log
for f in bench bench_again bench_fwd_bf16x3 bench_exstored bench_tn_bf16x3 bench_bwd_bf16x3 bench_eager bench_256 bench_256_eager bench_fp32_exact bench_mixed16 bench_shipped bench_blend bench_cfg5_bf16x3 bench_cfg5_mixed16 bench_cfg5_mixed16_exstored bench_strong4096; do python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,2), 'M rs/s', 'fwd', round(d.get('forward_only',{}).get('ms',0),3), d.get('roofline',{}).get('frac'))
except Exception as e: print('$f', 'ERR', e)
"; done
