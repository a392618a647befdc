"""Build libnudf.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m neuraludf_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box
with the gpurun snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libnudf.so")
SOURCES = ["nudf_api.hip", "gemm_f32_mfma.hip", "gemm_tn_f32_mfma.hip", "rays_embed.hip", "composite.hip", "upsample.hip",
           "blend.hip", "optim.hip", "mlp_chain.hip", "mlp_chain_rows.hip", "raybatch.hip"]


# the sources that decide what a kernel class reads and writes (bench.py `roofline.traffic_stale`: a PMC traffic file under
# profiles/ is only as good as the kernel it was recorded on)
KERNEL_SOURCES = {
    "mlp_chain": ["csrc/mlp_chain.hip", "csrc/mlp_chain_rows.hip", "csrc/mlp_chain_shared.h", "csrc/nudf_common.h",
                  "../include/nudf.h", "mlp.py"],
    "gemm_tn": ["csrc/gemm_tn_f32_mfma.hip", "csrc/nudf_common.h", "../include/nudf.h", "mlp.py"],
    "composite": ["csrc/composite.hip", "csrc/nudf_common.h", "../include/nudf.h"],
}


def source_digest(kernel: str) -> str:
    """sha256 (first 16 hex digits) over the source files of a kernel class, in a fixed order"""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES.get(kernel, sorted(os.path.join("csrc", f) for f in os.listdir(CSRC))):
        with open(os.path.join(HERE, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _flag_stamp() -> str:
    return LIB + ".flags"


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    # the extra compiler flags the library was built with (NUDF_HIPCC_FLAGS) are part of its identity
    stamp = open(_flag_stamp()).read() if os.path.exists(_flag_stamp()) else ""
    if stamp != os.environ.get("NUDF_HIPCC_FLAGS", ""):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "nudf.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def _obj_stale(obj: str, dep: str, flags: str) -> bool:
    """an object is reused when it is newer than every file its depfile (-MD) lists and was built with the same flags"""
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(obj + ".flags")):
        return True
    if open(obj + ".flags").read() != flags:
        return True
    t = os.path.getmtime(obj)
    txt = open(dep).read().replace("\\\n", " ")
    files = txt.split(":", 1)[1].split() if ":" in txt else []
    return any((not os.path.exists(f)) or os.path.getmtime(f) > t for f in files)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    procs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        d = o + ".d"
        objs.append(o)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o,
               "-I", CSRC, "-I", INCLUDE, "-Wno-unused-result"] + os.environ.get("NUDF_HIPCC_FLAGS", "").split()
        flags = " ".join(cmd)
        if not force and not _obj_stale(o, d, flags):
            continue
        for stale in (o, o + ".flags"):       # an interrupted / failed compile must not leave an object that looks current
            if os.path.exists(stale):
                os.remove(stale)
        cmd += ["-MD", "-MF", d]
        procs.append((cmd, flags, o, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = None
    for cmd, flags, o, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = failed or RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
            continue
        with open(o + ".flags", "w") as f:    # the stamp is written only once the object exists
            f.write(flags)
        if verbose and out:
            print(out.decode())
    if failed is not None:
        raise failed
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    with open(_flag_stamp(), "w") as f:
        f.write(os.environ.get("NUDF_HIPCC_FLAGS", ""))
    return LIB


if __name__ == "__main__":
    if "--digest" in sys.argv:
        print(source_digest(sys.argv[sys.argv.index("--digest") + 1]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
