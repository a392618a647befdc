"""micro-benchmark of nudf_gemm_nn / nudf_gemm_tn shapes (GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd import mlp

dev = torch.device("cuda:0")
res = {}
def bench(fn, flops, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    return us, flops / us / 1e6

from neuraludf_amd import _lib
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.lib().nudf_set_gemm_variant(VARIANT)
print("variant", VARIANT)
for (M, N, K) in [(65536, 256, 256), (8192, 256, 256), (32768, 256, 256), (65536, 128, 128), (65536, 256, 64)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev) * 0.05
    C1 = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev)
    X1 = torch.rand(M, N, device=dev); X2 = torch.rand(M, N, device=dev)
    bias = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    for epi, kw in [("NONE", dict(C1=C1)), ("SOFTPLUS", dict(C1=C1, C2=C2, bias=bias)), ("SOFTPLUS1", dict(C1=C1, bias=bias)),
                    ("MUL", dict(C1=C1, X1=X1)), ("TANGENT", dict(C1=C1, C2=C2, X1=X1, X2=X2)), ("BWD", dict(C1=C1, X1=X1, X2=X2))]:
        e = "SOFTPLUS" if epi == "SOFTPLUS1" else epi
        us, tf = bench(lambda: mlp.gemm_nn(A, B, M, N, K, e, **kw), fl)
        res[f"nn_{M}x{N}x{K}_{epi}"] = (round(us, 1), round(tf, 1))
for (M, NA, NB) in [(65536, 256, 256), (8192, 256, 256)]:
    A = torch.randn(M, NA, device=dev); B = torch.randn(M, NB, device=dev); C = torch.zeros(NA, NB, device=dev)
    us, tf = bench(lambda: mlp.gemm_tn(A, NA, B, C, NA, NB, M), 2.0 * M * NA * NB)
    res[f"tn_{M}x{NA}x{NB}"] = (round(us, 1), round(tf, 1))
    us, tf = bench(lambda: mlp.gemm_tn(A, NA, B, C, NA, NB, M, A2=A, na2=NA, B2=B), 4.0 * M * NA * NB)
    res[f"tn2_{M}x{NA}x{NB}"] = (round(us, 1), round(tf, 1))
for k, v in res.items():
    print(f"{k:40s} {v[0]:9.1f} us  {v[1]:7.1f} TF")
json.dump(res, open(f"gpurun_out/gemm_bench_v{VARIANT}.json", "w"))
