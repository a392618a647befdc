"""Is torch's CPU exp / sigmoid (Vectorized<float>: Sleef expf u10) reproduced bit for bit by the restatement that
`upsample_body.inc` carries under NUDF_UP_SLEEF?  numpy emulation with FMAs in extended precision against torch on this
host; run in the build container (AVX512) and, through gpurun, on the GPU box's host."""
import numpy as np
import torch

f32 = np.float32
LD = np.longdouble


def fma(a, b, c):
    # a*b exact in 64-bit-mantissa long double (24+24 bits), one rounding of the sum to 64 bits, then to fp32
    return (a.astype(LD) * b.astype(LD) + c.astype(LD)).astype(f32)


def sleef_expf_u10(d):
    d = d.astype(f32)
    R_LN2f = f32(1.442695040888963407359924681001892137426645954152985934135449406931)
    L2Uf, L2Lf = f32(0.693145751953125), f32(1.428606765330187045e-06)
    q = np.rint((d * R_LN2f).astype(f32)).astype(np.int32)
    qf = q.astype(f32)
    s = fma(qf, np.full_like(d, -L2Uf), d)
    s = fma(qf, np.full_like(d, -L2Lf), s)
    u = np.full_like(d, f32(0.000198527617612853646278381))
    for c in (0.00139304355252534151077271, 0.00833336077630519866943359, 0.0416664853692054748535156,
              0.166666671633720397949219, 0.5):
        u = fma(u, s, np.full_like(d, f32(c)))
    u = (f32(1.0) + fma((s * s).astype(f32), u, s)).astype(f32)
    # vldexp2: u * 2^(q>>1) * 2^(q - (q>>1))
    e1 = q >> 1
    e2 = q - e1
    p1 = ((e1 + 127).astype(np.int32) << 23).view(f32)
    p2 = ((e2 + 127).astype(np.int32) << 23).view(f32)
    u = ((u * p1).astype(f32) * p2).astype(f32)
    u = np.where(d < f32(-104), f32(0), u)
    u = np.where(d > f32(100), f32(np.inf), u)
    return u


def main():
    g = np.random.default_rng(0)
    x = np.concatenate([g.uniform(-30, 30, 2_000_000), g.normal(0, 2, 2_000_000), g.uniform(-90, 88, 500_000)]).astype(f32)
    ref = torch.exp(torch.from_numpy(x)).numpy()
    got = sleef_expf_u10(x)
    bad = ref.view(np.int32) != got.view(np.int32)
    print("exp: %d / %d differ" % (bad.sum(), x.size), torch.backends.cpu.get_cpu_capability())
    if bad.any():
        i = np.nonzero(bad)[0][:5]
        print(x[i], ref[i], got[i])
    # sigmoid = 1 / (1 + exp(-x)) with a true division
    refs = torch.sigmoid(torch.from_numpy(x)).numpy()
    gots = (f32(1) / (f32(1) + sleef_expf_u10(-x))).astype(f32)
    bads = refs.view(np.int32) != gots.view(np.int32)
    print("sigmoid: %d / %d differ" % (bads.sum(), x.size))
    # the short-tensor path (fewer elements than a vector): does torch use the same routine?
    xs = x[:5].copy()
    print("short exp equal:", np.array_equal(torch.exp(torch.from_numpy(xs)).numpy().view(np.int32), sleef_expf_u10(xs).view(np.int32)))
    # libm expf for comparison
    import math
    lib = np.array([np.float32(math.exp(float(v))) for v in x[:200000]], dtype=f32)   # correctly rounded double exp -> fp32
    print("vs correctly-rounded exp: torch differs on %.2f %%" % (100.0 * (lib.view(np.int32) != ref[:200000].view(np.int32)).mean()))


if __name__ == "__main__":
    main()
