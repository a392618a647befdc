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
    assert lib.nudf_version() >= 100
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
                      ("NudfAdamGroup", _lib.AdamGroup), ("NudfChainStep", _lib.ChainStep), ("NudfRayBatch", _lib.RayBatch)]:
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
