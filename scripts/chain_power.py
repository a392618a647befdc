"""are the chain launches clock / power limited too?  UDF forward + tangent sweep at 65 536 points (mlp_chain_kernel<64, 2>, two
launches) with the network's real (perturbed) weights on random points, and with every weight and bias zeroed (the same
instruction stream on operands that do not toggle): wall time per pair of launches and the card's power / clock (hwmon)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from neuraludf_amd import mlp
from neuraludf_amd.models import fields
from common import build_modules, perturb_
dev = torch.device("cuda:0")
M = 65536
for prec in ("bf16x3", "fp32"):
    mlp.set_precision(prec)
    for kind in ("trained-like weights, random points", "all weights and biases zero"):
        udf = perturb_(build_modules(fields, seed=0))["udf"].to(dev)
        if kind.startswith("all"):
            with torch.no_grad():
                for p in udf.parameters():
                    p.zero_()
        x = (torch.rand(M, 3, device=dev) * 2 - 1)

        def sweeps():
            with torch.no_grad():
                return udf.gradient(x)
        for _ in range(5):
            sweeps()
        torch.cuda.synchronize()
        with bench._PowerSampler(device=0) as ps:
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 1.2:
                for _ in range(20):
                    sweeps()
                torch.cuda.synchronize()
                n += 20
            dt = (time.perf_counter() - t0) / n
        pw = ps.summary() or {}
        print(f"{prec:>7}  {kind:<40} {dt * 1e3:.3f} ms per forward + tangent sweep   {pw.get('avg_w', 0):.0f} W  {pw.get('sclk_mhz_avg', 0):.0f} MHz")
