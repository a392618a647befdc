"""The 16-bit-TILE chain kernel (mlp_chain_kernel<64, 3>, BASELINE config 5's mode): the LDS activation tile holds the MFMA
operand type of the chain (fp16 forward sweeps / bf16 backward sweeps) instead of fp32 -- one ds_read_b128 per 32-row tile
and k step and no conversion in the K loop, 37 KB per 64-point workgroup (three per CU).  The activations are rounded to the
same type ONCE, in the producing epilogue, instead of in every reading wave: the MFMAs see the same operands in the same
order, so every output of every sweep must equal the fp32-tile 16-bit kernel's (mlp_chain_kernel<64, 1>) BIT FOR BIT.

Also measured here: what moving the abs-head column from the fp32 MFMA to fp16 operands (mlp.HEAD16, what lets the whole
forward sweep run on this kernel) does to the UDF value."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plain(t, P):
    from neuraludf_amd import mlp
    if t is None:
        return None
    if mlp._isp4(t):
        return mlp.unpack16(t)[:P]
    return t[:P] if t.dim() >= 1 and t.shape[0] >= P else t


def _all_sweeps(dev, P, seed):
    """forward with state, input gradient, second-order backward of the UDF network, the colour net's chains and the
    background NeRF's forward / backward -> tensors"""
    import chain_sweeps as CS
    from neuraludf_amd import mlp
    st_ = CS.engines(dev)
    eng, ceng = st_["eng"], st_["ceng"]
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    d_udf = torch.randn(P, generator=g).to(dev)
    d_g = torch.randn(P, 3, generator=g).to(dev)
    S = 64
    rays_d = torch.nn.functional.normalize(torch.randn((P + S - 1) // S, 3, generator=g), dim=-1).to(dev)
    out = {}
    st = eng.forward(x, need_grad_state=True, feat_ld=ceng.cin_ld)
    out.update(udf=st["udf"][:P], sign=st["sign"][:P], feat=st["feat"][:P, :256])
    for l in (1, 4, 5, 8):
        out[f"X{l}"] = _plain(st["X"][l], P)[:, :eng.layers[l].inp]
    gr, DA = eng.gradient(x, st)
    out["g"] = gr[:P]
    for l in (0, 3, 7):
        out[f"DA{l}"] = _plain(DA[l], P)[:, :eng.layers[l].out]       # (pad columns are never written)
    Pc = (P // S) * S
    cb, cc, logits, cst = ceng.forward(st["feat"], rays_d, S, Pc)
    out.update(cb=cb[:Pc], cc=cc[:Pc])
    d_cb = torch.randn(Pc, 3, generator=g).to(dev)
    d_cc = torch.randn(Pc, 3, generator=g).to(dev)
    d_lg = torch.randn(Pc, logits.shape[1], generator=g).to(dev) if logits is not None else None
    cgr, dCIN = ceng.backward(cst, cb, cc, d_cb, d_cc, d_lg)
    out["dCIN"] = dCIN[:Pc, :256]
    for i, t in enumerate(cgr):
        out[f"cg{i}"] = t
    d_feat = torch.zeros(P, ceng.cin_ld, device=dev)
    d_feat[:Pc] = dCIN[:Pc]
    grads = eng.backward(x, st, DA, d_udf, d_feat, ceng.cin_ld, d_g)
    for i, t in enumerate(grads):
        out[f"p{i}"] = t
    # the background NeRF's chains (ReLU layers, the 340-column skip layer through RELUADD, two positional encodings)
    nerf = st_["nerf"]
    Pn = (min(P, 32768) // S) * S
    pts4 = torch.randn(Pn, 4, generator=g).to(dev) * 0.5
    sig, rgb = nerf.evaluate(pts4, rays_d[:Pn // S].contiguous(), S)
    out.update(nsig=sig.detach(), nrgb=rgb.detach())
    (sig.sum() + (rgb * torch.randn(rgb.shape, generator=g).to(dev)).sum()).backward()
    for i, prm in enumerate(nerf.parameters()):
        if prm.grad is not None:
            out[f"n{i}"] = prm.grad.detach().clone()
            prm.grad = None
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in out.items() if v is not None}


@pytest.mark.parametrize("P", [64 * 300, 64 * 517 + 29])
def test_t16_kernel_equals_the_fp32_tile_kernel_bit_for_bit(P):
    import chain_sweeps as CS  # noqa: F401  (tests/ on sys.path)
    from neuraludf_amd import _lib, mlp
    dev = torch.device("cuda:0")
    base = mlp.PRECISION
    lib = _lib.lib()
    try:
        mlp.set_precision("mixed16")
        old = lib.nudf_set_chain_t16(1)
        a = _all_sweeps(dev, P, seed=5)
        lib.nudf_set_chain_t16(0)
        b = _all_sweeps(dev, P, seed=5)
    finally:
        lib.nudf_set_chain_t16(old)
        mlp.set_precision(base)
    assert set(a) == set(b) and len(a) > 40
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, [(k, float((a[k].float() - b[k].float()).abs().max())) for k in diff]
    assert all(bool(torch.isfinite(v.float()).all()) for v in a.values())
    print(f"16-bit-tile kernel vs fp32-tile 16-bit kernel at P = {P}: {len(a)} tensors bit-identical")


def test_fp16_head_against_the_fp32_head():
    """mlp.HEAD16: the abs-head column on fp16 operands (so that the forward sweep has one operand type) against the fp32
    head on the same fp16-rounded hidden activations -- the difference must stay at the level the mode's other roundings
    already put on the UDF value (reported; bar: 1e-3 of max |udf|, the colour PSNR bars are held by test_gpu_mixed16 /
    test_gpu_fullsize_parity)."""
    import chain_sweeps as CS
    from neuraludf_amd import mlp
    dev = torch.device("cuda:0")
    st_ = CS.engines(dev)
    eng = st_["eng"]
    P = 64 * 300
    x = (torch.rand(P, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1).to(dev)
    base, head = mlp.PRECISION, mlp.HEAD16
    res = {}
    try:
        mlp.set_precision("fp32")
        res["fp32"] = eng.forward(x, need_grad_state=False, udf_only=True)["udf"][:P].clone()
        mlp.set_precision("mixed16")
        for h in (False, True):
            mlp.HEAD16 = h
            eng.invalidate()
            res[h] = eng.forward(x, need_grad_state=False, udf_only=True)["udf"][:P].clone()
    finally:
        mlp.HEAD16 = head
        mlp.set_precision(base)
        eng.invalidate()
    ref = float(res["fp32"].abs().max())
    e32 = float((res[False] - res["fp32"]).abs().max()) / ref
    e16 = float((res[True] - res["fp32"]).abs().max()) / ref
    d = float((res[True] - res[False]).abs().max()) / ref
    print(f"mixed16 udf vs exact fp32 (max / max |udf|): fp32 head {e32:.2e}, fp16 head {e16:.2e}; head alone {d:.2e}")
    assert e16 < 2.0 * e32 + 2e-4, (e16, e32)
    assert d < 1e-3, d
