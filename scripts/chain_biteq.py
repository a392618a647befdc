import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from chain_sweeps import sweeps
dev = torch.device("cuda:0")
P = 64 * 300 + 21
a = sweeps(dev, P, 64, seed=8)
b = sweeps(dev, P, 66, seed=8)
same = [k for k in a if torch.equal(a[k], b[k])]
diff = [(k, float((a[k] - b[k]).abs().max())) for k in a if not torch.equal(a[k], b[k])]
print("bit-identical:", len(same), "of", len(a)); print("different:", diff[:30])
