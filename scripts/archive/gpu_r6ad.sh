#!/bin/bash
# round 6: per-wave chain timeline on the final build (the script now prices each step's MFMA cycles on the pipe it runs on)
mkdir -p gpurun_out/r6ad
timeout 300 python scripts/chain_timeline.py 65536 2>&1 | grep -v "Warn\|amdgpu.ids\|distinct" > gpurun_out/r6ad/chain_timeline.txt
head -8 gpurun_out/r6ad/chain_timeline.txt
