"""CPU checks of the C-ABI boundary: the library builds for gfx950, loads, exports every symbol
declared in include/nudf.h, and the product path refuses to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    from neuraludf_amd import build, _lib
    path = build.build()
    assert os.path.exists(path)
    lib = _lib.lib()
    assert lib.nudf_version() >= 105
    hdr = open(os.path.join(ROOT, "include", "nudf.h")).read()
    declared = sorted(set(re.findall(r"\b(nudf_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in include/nudf.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared


def test_ctypes_structs_match_header_field_counts():
    from neuraludf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nudf.h")).read()
    for cname, st in [("NudfGemmNN", _lib.GemmNN), ("NudfGemmTN", _lib.GemmTN), ("NudfComposite", _lib.Composite),
                      ("NudfCompositeGrad", _lib.CompositeGrad), ("NudfUpsample", _lib.Upsample),
                      ("NudfPixelBlend", _lib.PixelBlend), ("NudfPixelComposite", _lib.PixelComposite),
                      ("NudfPatchBlend", _lib.PatchBlend), ("NudfPatchWarp", _lib.PatchWarp), ("NudfAdamTensor", _lib.AdamTensor),
                      ("NudfAdamGroup", _lib.AdamGroup), ("NudfChainStep", _lib.ChainStep), ("NudfRayBatch", _lib.RayBatch),
                      ("NudfGemmTNProblem", _lib.GemmTNProblem), ("NudfBlendLoss", _lib.BlendLoss)]:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"[A-Za-z_0-9]+", part)[-1])
        assert names == [f[0] for f in st._fields_], cname


def test_ctypes_struct_layouts_match_a_c_compile_of_the_header(tmp_path):
    """sizeof and every field offset of the by-value argument structs, as gcc lays out include/nudf.h, against the ctypes
    mirrors of neuraludf_amd/_lib.py (the field-name test above cannot see padding or type widths)."""
    import ctypes as C
    import shutil
    import subprocess
    from neuraludf_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    pairs = [("NudfBlendLoss", _lib.BlendLoss), ("NudfGemmNN", _lib.GemmNN), ("NudfGemmTN", _lib.GemmTN), ("NudfGemmTNProblem", _lib.GemmTNProblem),
             ("NudfGemmTNGroup", _lib.GemmTNGroup), ("NudfComposite", _lib.Composite), ("NudfCompositeGrad", _lib.CompositeGrad),
             ("NudfUpsample", _lib.Upsample), ("NudfPixelBlend", _lib.PixelBlend), ("NudfPixelComposite", _lib.PixelComposite),
             ("NudfPatchBlend", _lib.PatchBlend), ("NudfPatchWarp", _lib.PatchWarp), ("NudfAdamTensor", _lib.AdamTensor),
             ("NudfAdamGroup", _lib.AdamGroup), ("NudfChainStep", _lib.ChainStep), ("NudfChain", _lib.Chain),
             ("NudfRayBatch", _lib.RayBatch), ("NudfAdam", _lib.Adam), ("NudfPackFrag", _lib.PackFrag),
             ("NudfPackLayer", _lib.PackLayer), ("NudfPackMulti", _lib.PackMulti), ("NudfUnpackLayer", _lib.UnpackLayer),
             ("NudfUnpackMulti", _lib.UnpackMulti)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "nudf.h"', 'int main(void) {']
    for cname, st in pairs:
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in st._fields_:
            cfield = "in" if f[0] == "in_" else f[0]        # `in` is a Python keyword: the mirrors call that field in_
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, cfield))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        a, b, c = ln.split()
        got[(a, b)] = int(c)
    for cname, st in pairs:
        assert got[(cname, "sizeof")] == C.sizeof(st), cname
        for f in st._fields_:
            assert got[(cname, f[0])] == getattr(st, f[0]).offset, (cname, f[0])


def test_weight_gradient_gemm_plan_host_logic():
    """the planner of nudf_gemm_tn_grouped is host code: workspace sizes (= workgroups x slot) of known groups, one
    resident wave of <= 512 workgroups, argument errors -- no GPU involved."""
    import ctypes as C
    from neuraludf_amd import _lib
    lib = _lib.lib()
    SLOT = 128 * 128 + 128

    def group(shapes, M, flags=0, ld=None):
        g = _lib.GemmTNGroup()
        g.n_problems, g.M, g.rows_per_block, g.prec = len(shapes), M, 0, 0
        for i, (NA, NB) in enumerate(shapes):
            q = g.prob[i]
            q.A1, q.B1, q.C = 4096, 8192, 16384          # fake 16-byte aligned addresses: never dereferenced here
            q.lda1, q.ldb1, q.ldc = ld or (NA + 3) // 4 * 4, ld or (NB + 3) // 4 * 4, NB
            q.NA, q.NB, q.flags = NA, NB, flags
        return g

    udf = [(256, 40)] + [(256, 256)] * 3 + [(217, 256)] + [(256, 256)] * 3 + [(256, 256), (1, 256)]
    n = lib.nudf_gemm_tn_grouped_workspace(C.byref(group(udf, 65536)))
    assert n % SLOT == 0 and 36 * 8 <= n // SLOT <= 512          # 36 tiles, several row chunks each, one resident wave
    small = lib.nudf_gemm_tn_grouped_workspace(C.byref(group([(3, 5)], 17)))
    assert small == SLOT                                         # one tile, one chunk (at least 8 k-steps per workgroup)
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(group([], 100))) == 0
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(group([(1024, 1024)] * 2, 100))) < 0     # 128 tiles > 64
    assert b"64 output tiles" in lib.nudf_last_error()
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(group([(64, 64)], 100, ld=66))) < 0       # ld % 4
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(group([(64, 64)], 100, flags=1 | 4))) < 0  # blocked is fp32-only
    # 16-bit launches and bf16 operands use whole quadrants and equal costs: still one resident wave
    g16 = group(udf, 262144, flags=3, ld=256)
    g16.prec = 2
    assert 0 < lib.nudf_gemm_tn_grouped_workspace(C.byref(g16)) // SLOT <= 512
    # with a workspace (fixed-order, non-atomic reduction) the problems of a group must write disjoint C / dbias ranges
    g2 = group([(64, 64), (64, 64)], 4096)
    g2.workspace = 1 << 20                                # fake non-NULL pointer: the planner never dereferences it
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(g2)) < 0           # both problems write C = 16384
    assert b"disjoint" in lib.nudf_last_error()
    g2.prob[1].C = 16384 + 4 * 64 * 64
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(g2)) > 0
    g2.prob[0].dbias, g2.prob[1].dbias = 1 << 16, (1 << 16) + 4 * 32       # [NA = 64] ranges overlap by 32 entries
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(g2)) < 0
    g2.prob[1].dbias = (1 << 16) + 4 * 64
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(g2)) > 0
    g2.workspace = None                                   # atomics path: overlapping outputs are allowed
    g2.prob[1].C = 16384
    assert lib.nudf_gemm_tn_grouped_workspace(C.byref(g2)) > 0


def test_per_layer_weight_gradient_gemm_refuses_formatted_operands():
    """nudf_gemm_tn reads row-major fp32 only: the per-layer fallback must refuse bf16-stored / blocked-layout state
    instead of producing silent garbage."""
    from neuraludf_amd import mlp
    from neuraludf_amd._lib import NudfError
    a = torch.zeros(64, 32)
    with pytest.raises(NudfError, match="gemm_tn_grouped"):
        mlp.gemm_tn(a.to(torch.bfloat16), 32, a, torch.zeros(32, 32), 32, 32, 64)
    with pytest.raises(NudfError, match="gemm_tn_grouped"):
        mlp.gemm_tn(a, 32, mlp.block(a), torch.zeros(32, 32), 32, 32, 64)


def test_weight_gradient_gemm_workgroup_order_host_logic():
    """blockIdx -> (tile, row chunk) of nudf_gemm_tn_grouped (host mirror of the kernels' decode): the live workgroups are a
    bijection onto the workspace slots, the tiles of a problem that read the same row chunk share blockIdx % 8 = the XCD --
    EVERY chunk (round 5: the last, narrower column of eight keeps its lanes and leaves holes, which exit) -- and no XCD
    receives more live workgroups than it has slots (2 per CU x 32 CUs)."""
    import collections
    import ctypes as C
    from neuraludf_amd import _lib
    lib = _lib.lib()

    def plan(shapes, M, flags=0, prec=0, ld=None):
        g = _lib.GemmTNGroup()
        g.n_problems, g.M, g.rows_per_block, g.prec = len(shapes), M, 0, prec
        for i, (NA, NB) in enumerate(shapes):
            q = g.prob[i]
            q.A1, q.B1, q.C = 4096, 8192, 16384
            q.lda1, q.ldb1, q.ldc = ld or (NA + 3) // 4 * 4, ld or (NB + 3) // 4 * 4, NB
            q.NA, q.NB, q.flags = NA, NB, flags
        out = (C.c_int32 * (4 * 2048))()
        n = lib.nudf_gemm_tn_grouped_plan(C.byref(g), out, 2048)
        assert 0 < n <= 2048
        assert lib.nudf_gemm_tn_grouped_workspace(C.byref(g)) % sum(1 for b in range(n) if out[4 * b] >= 0) == 0
        return [tuple(out[4 * b:4 * b + 4]) for b in range(n)]

    udf = [(256, 40)] + [(256, 256)] * 3 + [(217, 256)] + [(256, 256)] * 3 + [(256, 256), (1, 256)]
    color = [(256, 295)] + [(256, 256)] * 3 + [(3, 256)]
    for shapes, M, kw in [(udf, 65536, {}), (udf, 65536, dict(prec=3)), (color, 65536, dict(prec=3)), (udf, 5000, {}),
                          ([(129, 72), (3, 128), (256, 256)], 777, {}), (udf, 262144, dict(flags=3, prec=2, ld=256)),
                          ([(256, 256)], 65536, dict(prec=3)), ([(512, 384)] * 2, 40000, dict(prec=3))]:
        grid = plan(shapes, M, **kw)
        live = [(bid, b) for bid, b in enumerate(grid) if b[0] >= 0]
        assert all(b == (-1, -1, -1, -1) for b in grid if b[0] < 0)
        assert len(live) <= 512 and len(grid) <= len(live) * 1.5 + 64
        assert sorted(b[3] for _, b in live) == list(range(len(live)))                      # every slot exactly once
        per_chunk = collections.defaultdict(set)
        chunks_of = collections.Counter((b[0], b[1]) for _, b in live)
        for bid, (prob, tile, chunk, _) in live:
            per_chunk[(prob, chunks_of[(prob, tile)], chunk)].add((tile, bid % 8))
        shared = 0
        for (prob, n_chunks, chunk), members in per_chunk.items():
            if len(members) > 1:
                assert len({x for _, x in members}) == 1, (prob, chunk, members)
                shared += 1
        if M >= 5000 and len(shapes) > 1:
            assert shared > 0
        per_xcd = collections.Counter(bid % 8 for bid, _ in live)
        assert max(per_xcd.values()) <= 64, per_xcd
        if len(live) > 256:
            assert min(per_xcd.values()) >= 0.75 * max(per_xcd.values()), per_xcd


def test_blocked_layout_helpers_are_inverse():
    from neuraludf_amd import mlp
    t = torch.arange(128 * 24, dtype=torch.float32).reshape(128, 24)
    b = mlp.block(t)
    assert mlp._isblk(b) and not mlp._isblk(t) and b.shape == t.shape
    r, c = 77, 13                                                   # the formula of include/nudf.h
    assert float(b.reshape(-1)[(r // 32) * 32 * 24 + (c // 4) * 128 + (r % 32) * 4 + c % 4]) == float(t[r, c])
    assert torch.equal(mlp.unblock(b), t)
    assert mlp.unblock(t) is t


def test_no_cpu_fallback():
    from neuraludf_amd._lib import NudfError
    from neuraludf_amd.train import Trainer
    from neuraludf_amd import synth
    tr = Trainer(torch.device("cpu"), dict(n_samples=16, n_importance=0, n_outside=0, up_sample_steps=1, perturb=0.0))
    rays = synth.make_rays(synth.make_scene("tiny"), 0, 4)
    with pytest.raises(NudfError):
        tr.loss(rays)


def test_product_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "neuraludf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle", src, re.M):
                    bad.append(f)
    assert not bad, bad
