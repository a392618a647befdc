"""Probe (one GPU, one rank): can an RCCL all-reduce be captured in a HIP graph and replayed in this image?  Decides whether
train.GraphedStep may be enabled for data-parallel trainers (its step holds two all-reduces)."""
import os
import socket
import sys

import torch
import torch.distributed as dist

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(port))
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    x = torch.ones(1024, device=dev)
    dist.all_reduce(x)                       # eager warm-up (communicator set-up)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    y = torch.full((1024,), 2.0, device=dev)
    with torch.cuda.graph(g):
        z = y * 3.0
        dist.all_reduce(z)
        w = z + 1.0
    for i in range(3):
        y.fill_(float(i))
        g.replay()
        torch.cuda.synchronize()
        assert float(w[0]) == 3.0 * i + 1.0, (i, float(w[0]))
    print("RCCL all-reduce captured and replayed in a HIP graph: OK")
except Exception as e:                       # noqa: BLE001
    print("RCCL capture FAILED:", repr(e))
    sys.exit(0)
finally:
    dist.destroy_process_group()
