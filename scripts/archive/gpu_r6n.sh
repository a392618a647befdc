#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rays in 256 512; do for fast in 0 1; do echo -n "rays $rays NUDF_HOST_FAST=$fast: "; NUDF_HOST_FAST=$fast timeout 300 python scripts/host_profile.py $rays 2>&1 | grep "host enqueue"; done; done
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_graph.py tests/test_gpu_train_parity.py tests/test_gpu_dist.py tests/test_gpu_bucket_alias.py tests/test_gpu_optim.py tests/test_gpu_kernels.py tests/test_gpu_bf16x3.py tests/test_gpu_tn_plan.py tests/test_gpu_mixed16.py tests/test_gpu_blend.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error" | cut -c1-300
