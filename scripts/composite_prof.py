import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
r = bench.composite_roofline(dev, n, 256, reps=5)
