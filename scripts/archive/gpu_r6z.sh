#!/bin/bash
# is the f16x2-backward timing probe training on sensible numbers?  (loss seed 2^24 in both legs)
mkdir -p gpurun_out/r6z
NUDF_PROBE_SEED_LOG2=24 timeout 900 python scripts/debug_probe_bwd.py 2>&1 | grep -v Warn | tail -8 | tee gpurun_out/r6z/probe_check.txt
