#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_chain_t16.py tests/test_gpu_mixed16.py -q -s --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; grep -E "16-bit-tile|passed|failed|^E " $O/pytest.log | head -20
NUDF_PRECISION=mixed16 timeout 300 python - <<'PY' 2>&1 | grep -v Warning | tail -5
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch
import chain_sweeps as CS
from neuraludf_amd import mlp
dev = torch.device("cuda:0")
mlp.PROFILE = []
CS.sweeps(dev, 65536, 0, seed=1)
torch.cuda.synchronize()
for name, fl, s, e, detail, nb in mlp.PROFILE:
    if name == "mlp_chain":
        print(detail, round(s.elapsed_time(e) * 1e3, 1), "us")
PY
