#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
timeout 300 python scripts/host_profile.py 256 > $O/host_profile.txt 2>&1; grep -v Warning $O/host_profile.txt | head -60
