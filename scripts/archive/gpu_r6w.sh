#!/bin/bash
# round 6: why the colour-weight-ramp test sees no difference with f16x2 weight-gradient GEMMs
mkdir -p gpurun_out/r6w
timeout 600 python scripts/debug_ramp.py > gpurun_out/r6w/ramp.log 2>&1
tail -40 gpurun_out/r6w/ramp.log
