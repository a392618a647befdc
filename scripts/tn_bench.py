"""time nudf_gemm_tn on the UDF weight-gradient shape (two operand pairs, M points)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd import mlp
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for (NA, NB, two) in [(256, 256, True), (256, 256, False), (128, 128, False), (217, 256, True)]:
    A1 = torch.randn(M, mlp.pad32(NA), device=dev); B1 = torch.randn(M, mlp.pad32(NB), device=dev)
    A2 = torch.randn(M, mlp.pad32(NA), device=dev); B2 = torch.randn(M, mlp.pad32(NB), device=dev)
    C = torch.zeros(mlp.pad32(NA), mlp.pad32(NB), device=dev); db = torch.zeros(NA, device=dev)
    def run():
        if two: mlp.gemm_tn(A1, NA, B1, C, NA, NB, M, dbias=db, A2=A2, na2=NA, B2=B2)
        else: mlp.gemm_tn(A1, NA, B1, C, NA, NB, M, dbias=db)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * M * NA * NB * (2 if two else 1)
    print(f"M={M} NA={NA} NB={NB} pairs={2 if two else 1}: {us:.1f} us  {fl / us / 1e6:.1f} TF")
