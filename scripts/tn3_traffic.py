"""launches for the PMC passes of scripts/pmc_tn3.sh: the bf16x3 weight-gradient kernel on (a) ONE 128 x 128 tile (no operand
sharing possible: the byte count that calibrates FETCH_SIZE for this kernel's 4-byte row requests), (b) a 256 x 128 and a
256 x 256 problem (tiles sharing panels), (c) the UDF adjoint group.  Prints the algorithmic operand bytes of each launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd import mlp
dev = torch.device("cuda:0")
M = 65536
GROUPS = {
    "one_tile": [(128, 128)],
    "two_tiles": [(256, 128)],
    "four_tiles": [(256, 256)],
    "udf_adjoint": [(256, 40)] + [(256, 256)] * 3 + [(217, 256)] + [(256, 256)] * 3 + [(256, 256), (1, 256)],
}
mlp.set_precision("bf16x3")
torch.manual_seed(0)
for name in sys.argv[1:] or list(GROUPS):
    jobs = []
    nbytes = 0
    for NA, NB in GROUPS[name]:
        lda = max(4, (NA + 3) // 4 * 4)
        jobs.append((torch.randn(M, lda, device=dev), NA, torch.randn(M, NB, device=dev), NB,
                     torch.zeros((NA + 31) // 32 * 32, NB, device=dev), torch.zeros((NA + 31) // 32 * 32, device=dev)))
        nbytes += M * (lda + NB) * 4
    for _ in range(4):
        mlp.gemm_tn_grouped(jobs, M)
    torch.cuda.synchronize()
    print(f"{name}: 4 launches, operand bytes once each {nbytes / 1e6:.1f} MB")
