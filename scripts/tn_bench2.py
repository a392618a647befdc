import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd import mlp
from neuraludf_amd._lib import GemmTN, call, ptr
dev = torch.device("cuda:0")
M = 65536
NA = NB = 256
A1 = torch.randn(M, NA, device=dev); B1 = torch.randn(M, NB, device=dev)
A2 = torch.randn(M, NA, device=dev); B2 = torch.randn(M, NB, device=dev)
C = torch.zeros(NA, NB, device=dev); db = torch.zeros(NA, device=dev)
for rpb in (256, 512, 1024, 2048, 4096):
    a = GemmTN()
    a.A1, a.lda1, a.na1, a.B1, a.ldb1 = ptr(A1), NA, NA, ptr(B1), NB
    a.A2, a.lda2, a.na2, a.B2, a.ldb2 = ptr(A2), NA, NA, ptr(B2), NB
    a.C, a.ldc, a.dbias, a.M, a.NA, a.NB, a.rows_per_block = ptr(C), NB, ptr(db), M, NA, NB, rpb
    call("nudf_gemm_tn", a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call("nudf_gemm_tn", a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"rows_per_block={rpb}: {us:.1f} us {2.0 * 2 * M * NA * NB / us / 1e6:.1f} TF blocks={4 * ((M + rpb - 1) // rpb)}")
