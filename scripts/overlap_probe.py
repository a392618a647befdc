"""what would running a weight-gradient GEMM group BESIDE a chain sweep buy (two streams) -- the GEMM is bound by the matrix pipe /
the power limit, the tangent sweep mostly by HBM?  Sequential vs concurrent wall time of {UDF forward + tangent sweep at 65 536
points} and {two UDF weight-gradient groups}, random weights / activations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from neuraludf_amd import mlp
from neuraludf_amd.models import fields
from common import build_modules, perturb_
dev = torch.device("cuda:0")
mlp.set_precision("bf16x3")
M = 65536
udf = perturb_(build_modules(fields, seed=0))["udf"].to(dev)
x = (torch.rand(M, 3, device=dev) * 2 - 1)
SHAPES = [(256, 40)] + [(256, 256)] * 3 + [(217, 256)] + [(256, 256)] * 3 + [(256, 256), (1, 256)]


def make_jobs():
    jobs = []
    for NA, NB in SHAPES:
        lda = max(4, (NA + 3) // 4 * 4)
        jobs.append((torch.nn.functional.softplus(torch.randn(M, lda, device=dev)), NA, torch.randn(M, NB, device=dev) * 1e-3, NB,
                     torch.zeros((NA + 31) // 32 * 32, NB, device=dev), torch.zeros((NA + 31) // 32 * 32, device=dev)))
    return jobs


jobs_a, jobs_b = make_jobs(), make_jobs()


def sweeps():
    with torch.no_grad():
        return udf.gradient(x)


def gemms():
    mlp.gemm_tn_grouped(jobs_a, M)
    mlp.gemm_tn_grouped(jobs_b, M)


side = torch.cuda.Stream()
for _ in range(3):
    sweeps(); gemms()
torch.cuda.synchronize()


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def seq():
    sweeps(); gemms()


def conc():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gemms()
    sweeps()
    torch.cuda.current_stream().wait_stream(side)


def conc2():            # the GEMMs first in the main queue, the sweeps on the side stream
    side.wait_stream(torch.cuda.current_stream())
    gemms()
    with torch.cuda.stream(side):
        sweeps()
    torch.cuda.current_stream().wait_stream(side)


for rep in range(2):
    print(f"sweeps alone {timed(sweeps):.3f} ms, gemms alone {timed(gemms):.3f} ms, sequential {timed(seq):.3f} ms, "
          f"concurrent (gemms on the side stream) {timed(conc):.3f} ms, concurrent (sweeps on the side stream) {timed(conc2):.3f} ms")
