"""sha256 of every output of tests/chain_sweeps.py (all chain launches of a train step's MLP work) in the current build and
precision: run under two builds (NUDF_LIB=...) and diff the output to show a kernel change is bit-neutral."""
import sys, os, hashlib
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
from chain_sweeps import sweeps
dev = torch.device("cuda:0")
for P in (64 * 300 + 21, 8192):
    a = sweeps(dev, P, 0, seed=8)
    torch.cuda.synchronize()
    for k in sorted(a):
        print(P, k, hashlib.sha256(a[k].detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16])
