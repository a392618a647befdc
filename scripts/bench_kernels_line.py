"""print the per-kernel lines of a bench.py JSON line (stdin)"""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    k = d.get("kernels", {})
    print("ms/step %.3f  fwd_only %.3f ms | chain %.3f ms %.1f TF | gemm_tn %.3f ms %.1f TF" % (
        d["ms_per_step"], d["forward_only"]["ms"], k["mlp_chain"]["ms"], k["mlp_chain"]["tflops"], k["gemm_tn"]["ms"], k["gemm_tn"]["tflops"]))
