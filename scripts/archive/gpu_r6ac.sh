#!/bin/bash
# round 6: exact-scaling test over all reverse chains; what the f16x2 scalings add to the eager host cost
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6ac; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|pytest exit|^E  " $O/pytest.log | cut -c1-300
for rep in 1 2; do for sw in 0 1; do echo -n "rays 512 NUDF_TN_F16X2=NUDF_BWD_F16X2=$sw: "; NUDF_TN_F16X2=$sw NUDF_BWD_F16X2=$sw timeout 300 python scripts/host_profile.py 512 2>&1 | grep "host enqueue"; done; done | tee $O/host.txt
