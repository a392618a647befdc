#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
timeout 300 python scripts/tnw_phases.py 65536 > $O/tnw_phases.txt 2>&1; tail -6 $O/tnw_phases.txt
