"""Host-side sequencing of the MLP GEMM chains on libnudf (no torch arithmetic on activations).

Three engines, each a pair of (forward, hand-derived backward).  The default path of every engine is the FUSED LAYER CHAIN
(`nudf_mlp_chain`: all layers of a sweep in one launch, activations resident in LDS, csrc/mlp_chain.hip) plus the grouped
weight-gradient GEMM (`nudf_gemm_tn_grouped`); the per-layer launches over `nudf_gemm_nn / nudf_gemm_tn` remain as the
cross-check path (`USE_CHAIN = False`).  MFMA operand precision: `PRECISION` ("bf16x3" by default = fp32 products emulated on
the bf16 matrix pipe, "fp32" = exact fp32 MFMA, "mixed16" = BASELINE config 5):
  * UDFEngine    -- UDFNetwork.forward + .gradient (models/fields.py:192-231) and their
                    backward, including the second-order part (the reference differentiates
                    through autograd.grad(create_graph=True));
  * ColorEngine  -- ResidualRenderingNetwork.forward (fields.py:452-495);
  * NerfEngine   -- NeRF.forward with view directions (fields.py:599-628).
torch is used for buffer allocation (caching allocator), streams and autograd plumbing.

Activation buffers are [P, ld] with ld = roundup(width, 32); pad columns are zero.  Weights are
re-normalised (weight_norm) and packed by `nudf_weightnorm_pack` into W [out_pad, in_pad] and
W^T [in_pad, out_pad]; packing is cached on the parameters' version counters.
"""
from __future__ import annotations

import math
from typing import List, Optional

import ctypes as C

import torch

from . import _lib
import os

from ._lib import CH, CH_INIT, CH_MAX_STEPS, EPI, Chain, GemmNN, GemmTN, call, ptr


def pad32(n: int) -> int:
    return (n + 31) // 32 * 32


def pad_rows(P: int) -> int:
    """the fused chain kernels store whole 64-point tiles: their buffers carry roundup(P, 64) rows."""
    return (P + 63) // 64 * 64


def _buf(P, width, dev, zero=None, dtype=torch.float32, blocked=False):
    """[pad_rows(P), pad32(width)] fp32 (rows >= P are scratch for the tile kernels); pad columns zeroed (default),
    everything zeroed (zero=True) or nothing (zero=False).  dtype = bfloat16: stored state of the 16-bit mode, which is
    4-POINT PACKED (include/nudf.h, NUDF_CH_STATE16: element (r, c) at ((r // 4) ld + c) 4 + r % 4; `unpack16` gives the
    plain view) -- only the chain kernels and the grouped weight-gradient GEMM address it.
    blocked = True marks the buffer as holding the BLOCKED layout of include/nudf.h (same size; only the
    transposed-product chain kernel and the grouped weight-gradient GEMM address it; `unblock` gives the plain view)."""
    if blocked:
        assert zero is False and dtype == torch.float32
        t = torch.empty((pad_rows(P), pad32(width)), device=dev, dtype=dtype)
        t._nudf_blk = True
        return t
    ld = pad32(width)
    if dtype == torch.bfloat16:
        # pad columns: zero bits are zero in the packed layout as well -- the whole buffer is cleared only when asked for
        t = (torch.zeros if zero else torch.empty)((pad_rows(P), ld), device=dev, dtype=dtype)
        if zero is None and ld != width:
            t.view(pad_rows(P) // 4, ld, 4)[:, width:, :].zero_()
        t._nudf_p4 = True
        return t
    if zero is None:          # only the pad COLUMNS must be finite zeros (they meet zero weight rows / unread dW columns);
        t = torch.empty((pad_rows(P), ld), device=dev, dtype=dtype)            # a full fill of a [65536, 224] buffer
        if ld != width:                                                        # costs 15 us, the 7 pad columns 3 us
            t[:, width:].zero_()
        return t
    return (torch.zeros if zero else torch.empty)((pad_rows(P), ld), device=dev, dtype=dtype)


def _is16(t):
    return t is not None and t.dtype == torch.bfloat16


def _isblk(t):
    return t is not None and getattr(t, "_nudf_blk", False)


def _isp4(t):
    return t is not None and getattr(t, "_nudf_p4", False)


def pack16(t):
    """4-point packed bf16 copy of a row-major [R, ld] buffer (R % 4 == 0): the 16-bit stored-state layout."""
    R, ld = t.shape
    assert R % 4 == 0
    b = t.to(torch.bfloat16).reshape(R // 4, 4, ld).permute(0, 2, 1).reshape(R, ld).contiguous()
    b._nudf_p4 = True
    return b


def unpack16(t):
    """row-major fp32 copy of a 4-point packed bf16 buffer; inverse of `pack16`."""
    R, ld = t.shape
    assert _isp4(t) and R % 4 == 0
    return t.reshape(R // 4, ld, 4).permute(0, 2, 1).reshape(R, ld).float()


def block(t):
    """blocked-layout copy of a row-major [R, ld] fp32 buffer (R % 32 == 0, ld % 4 == 0); inverse of `unblock`."""
    R, ld = t.shape
    assert R % 32 == 0 and ld % 4 == 0 and t.dtype == torch.float32
    b = t.reshape(R // 32, 32, ld // 4, 4).permute(0, 2, 1, 3).reshape(R, ld).contiguous()
    b._nudf_blk = True
    return b


def unblock(t):
    """row-major copy of a buffer in the blocked layout (tests / debugging): element (r, c) sits at
    (r // 32) * 32 * ld + (c // 4) * 128 + (r % 32) * 4 + c % 4."""
    if not _isblk(t):
        return t
    R, ld = t.shape
    return t.reshape(R // 32, ld // 4, 32, 4).permute(0, 2, 1, 3).reshape(R, ld).contiguous()


def _zero_cols(t, c0):
    """zero the pad columns [c0, ld) of a [P, ld] buffer (they multiply zero weight rows, so they must be finite)."""
    if c0 < t.shape[1]:
        t[:, c0:].zero_()
    return t


def alloc_grads(layers, zero=True):
    """one flat buffer for the packed weight / bias gradients of `layers` -> [(dW, db), ...] views.  zero = False: the
    first weight-gradient launch ASSIGNS every element the unpack kernel reads (`gemm_tn_grouped(..., assign=True)`), so
    the fill (a 5 us launch per backward) is skipped; pad rows / columns then hold arbitrary values nobody reads."""
    sizes = [(pl.out_pad * pl.in_pad, pl.out_pad) for pl in layers]     # db padded to out_pad keeps 16-byte alignment
    flat = (torch.zeros if zero else torch.empty)(sum(a + b for a, b in sizes), device=layers[0].W.device)
    out, off = [], 0
    for pl, (a, b) in zip(layers, sizes):
        dW = flat[off:off + a].view(pl.out_pad, pl.in_pad)
        db = flat[off + a:off + a + pl.out]
        out.append((dW, db))
        off += a + b
    return out


def gemm_nn(A, B, M, N, K, epi, C1=None, ldc1=0, C2=None, ldc2=0, C3=None, ldc3=0, X1=None, ldx1=0, X2=None, ldx2=0,
            bias=None, lda=None, ldb=None, iparam=0, scale=1.0, xscale=1.0, c1_off=0, c2_off=0, x1_off=0, x2_off=0):
    """thin wrapper filling NudfGemmNN; *_off are column offsets (in floats) into the buffers."""
    a = GemmNN()
    a.A, a.lda = ptr(A), (lda if lda is not None else A.shape[1])
    a.B, a.ldb = ptr(B), (ldb if ldb is not None else B.shape[1])
    a.bias = ptr(bias)
    a.C1 = (ptr(C1) + 4 * c1_off) if C1 is not None else None
    a.ldc1 = ldc1 or (C1.shape[1] if (C1 is not None and C1.dim() == 2) else 0)
    a.C2 = (ptr(C2) + 4 * c2_off) if C2 is not None else None
    a.ldc2 = ldc2 or (C2.shape[1] if (C2 is not None and C2.dim() == 2) else 0)
    a.C3 = ptr(C3)
    a.ldc3 = ldc3 or (C3.shape[1] if (C3 is not None and C3.dim() == 2) else 0)
    a.X1 = (ptr(X1) + 4 * x1_off) if X1 is not None else None
    a.ldx1 = ldx1 or (X1.shape[1] if (X1 is not None and X1.dim() == 2) else 0)
    a.X2 = (ptr(X2) + 4 * x2_off) if X2 is not None else None
    a.ldx2 = ldx2 or (X2.shape[1] if (X2 is not None and X2.dim() == 2) else 0)
    a.M, a.N, a.K, a.epi, a.iparam, a.scale, a.xscale = M, N, K, EPI[epi], iparam, scale, xscale
    if PROFILE is not None:
        _timed("gemm_nn", 2.0 * M * N * getattr(B, "k_true", K), lambda: call("nudf_gemm_nn", a))
        return
    call("nudf_gemm_nn", a)


# MFMA products a chain step executes per fp32 product, by NudfChainStep.prec (0 fp32, 1 fp16, 2 bf16, 3 bf16x3, 4 f16x2)
MFMA_PRODUCTS = {0: 1, 1: 1, 2: 1, 3: 6, 4: 3}

# bench.py sets PROFILE = [] for one instrumented step: every GEMM launch is bracketed by HIP events on
# the launch stream and recorded as (kernel, algorithmic flops, start, end)
PROFILE = None


def _timed(name, flops, fn, detail=None, nbytes=0.0, xflops=None):
    """`detail`: which instantiation / sweep the launch is (bench.py's roofline.per_kernel); `nbytes`: the launch's
    ALGORITHMIC HBM bytes (stored-state arrays it must read and write once), for the HBM side of its roofline; `xflops`: the
    flops the launch EXECUTES on the matrix pipe (split modes: 6 or 3 MFMA products per fp32 product; None: the class's factor)."""
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    PROFILE.append((name, flops, s, e, detail or name, nbytes, xflops))


def call_timed(cls, detail, nbytes, name, *args, units=0.0):
    """`call(name, *args)`; during bench.py's instrumented step (PROFILE is a list) bracketed by HIP events and recorded as
    a launch of the non-MFMA class `cls` with its ALGORITHMIC bytes (`units`: taps / samples, carried in the flops slot with
    a negative sign so that the MFMA accounting skips it)."""
    if PROFILE is None:
        call(name, *args)
    else:
        _timed(cls, -float(units), lambda: call(name, *args), detail, float(nbytes))


def gemm_tn(A1, na1, B1, C, NA, NB, M, dbias=None, A2=None, na2=0, B2=None):
    # nudf_gemm_tn reads its operands as row-major fp32: bf16-stored (mixed16 state) or blocked-layout buffers carry their
    # format only through gemm_tn_grouped's per-problem flags
    for t in (A1, B1, A2, B2):
        if t is not None and (_is16(t) or _isblk(t)):
            raise _lib.NudfError("gemm_tn: bf16-stored / blocked-layout operands need gemm_tn_grouped (per-problem "
                                 "NUDF_TN_A16 / _B16 / _A_BLK / _B_BLK flags); set NUDF_STATE16=0 / NUDF_BLOCKED_STATE=0 "
                                 "with NUDF_UDF_TN_GROUPED=0")
    a = GemmTN()
    a.A1, a.lda1, a.na1 = ptr(A1), A1.shape[1], na1
    a.B1, a.ldb1 = ptr(B1), B1.shape[1]
    if A2 is not None:
        a.A2, a.lda2, a.na2 = ptr(A2), A2.shape[1], na2
        a.B2, a.ldb2 = ptr(B2), B2.shape[1]
    a.C, a.ldc = ptr(C), C.shape[1]
    a.dbias = ptr(dbias)
    a.M, a.NA, a.NB, a.rows_per_block = M, NA, NB, 0
    a.prec = _tn_prec()
    if PROFILE is not None:
        _timed("gemm_tn", 2.0 * M * NA * NB * (2 if A2 is not None else 1), lambda: call("nudf_gemm_tn", a))
        return
    call("nudf_gemm_tn", a)


# ------------------------------------------------------------------------------------------
# fused layer chains (csrc/mlp_chain.hip)
# ------------------------------------------------------------------------------------------
USE_CHAIN = os.environ.get("NUDF_CHAIN", "1") != "0"     # 0: per-layer GEMM launches (A/B measurements, cross-checks)
CHAIN_DEBUG = None     # int64 tensor [blocks * 4, 32]: per-wave timeline written by the kernel (scripts/chain_timeline.py)
CHAIN_TILE = int(os.environ.get("NUDF_CHAIN_TILE", "0"))  # 0 = auto, 32 / 64 force the points-per-workgroup tile
# the colour net's three chains alone ("" = as CHAIN_TILE; A/B switch: its 128-wide layers behave differently from the UDF net's)
COLOR_TILE = int(os.environ.get("NUDF_COLOR_TILE", "0"))


def k8(n: int) -> int:
    """reduction length of a chain step: the K loop runs two groups of 8 per iteration."""
    return (n + 15) // 16 * 16


def pack_frag(B, ldb, K, N, off=0):
    """[K, N] operand (row stride ldb, starting `off` floats into B) -> MFMA-B fragment order."""
    out = torch.empty(k8(K) // 8 * ((N + 31) // 32) * 256, device=B.device, dtype=torch.float32)
    call("nudf_pack_frag", ptr(B) + 4 * off, ldb, K, N, ptr(out))
    out.k_true, out.n_true = K, N
    return out


_PLAIN_TYPES = frozenset((int, float, str, bool, type(None)))
# filled chain descriptors of the previous launch per call site (ChainBuilder(site=...)): eager host cost, DESIGN section 6
_CHAIN_MEMO = {}
CHAIN_MEMO = os.environ.get("NUDF_CHAIN_MEMO", "1") != "0" and _lib.HOST_FAST
chain_memo_hits = 0


def _memo_token(engine):
    """identity of a call site's owner in the descriptor memo: an object the engine owns (id() of a collected engine could be
    reused by another one while its entries are still in the memo; an object held by the key cannot)"""
    t = engine.__dict__.get("_site_token")
    if t is None:
        t = engine.__dict__["_site_token"] = object()
    return t


class ChainBuilder:
    """fills a NudfChain; keeps the tensors it points at alive until the launch is enqueued.

    The public calls (posenc / init_* / step) only RECORD their arguments; `launch()` fills the 2 KB descriptor from the
    record -- or, for a builder that names its call `site`, reuses the descriptor the same site filled last time when the
    record is the same launch again: same sizes, scalars and operand mode, every tensor at the same address with the same
    element count (the steady state of a training loop, where torch's caching allocator hands every call site the same blocks
    step after step).  Filling costs ~9 us of Python per step of a chain, 0.9 ms per train step; comparing the record ~1.5 us."""

    def __init__(self, P, init, k0, tile_rows=0, site=None):
        self.P, self.init, self.k0, self.tile_rows = P, init, k0, (tile_rows or CHAIN_TILE)
        self.site = site
        self.rec = []
        self.c = None
        self.n = 0
        self.flops = 0.0
        self.xflops = 0.0
        self.nbytes = 0.0          # algorithmic HBM bytes: every stored-state operand / output of every step, once
        self.epis = []
        self.blocked = False
        self.keep = []

    # ---- recording front end -------------------------------------------------------------------------------------------
    def posenc(self, *a, **kw):
        self.rec.append((0, a, kw))

    def init_store(self, *a, **kw):
        self.rec.append((1, a, kw))

    def init_load(self, *a, **kw):
        self.rec.append((2, a, kw))

    def init_seed(self, *a, **kw):
        self.rec.append((3, a, kw))

    def step(self, *a, **kw):
        self.rec.append((4, a, kw))

    def absmax(self, *a, **kw):
        """one zeroed device float the launch raises to the largest |value| it stores for a weight-gradient GEMM to read
        (NudfChain.absmax_out): the scale of that side of an f16x2 GEMM"""
        self.rec.append((5, a, kw))

    def tile_scale(self, *a, **kw):
        """(amax_in=None, amax_out=None): run this LINEAR sweep with every tile of points multiplied by its own power of two
        (NudfChain.tile_scale) -- what lets a backward sweep, whose operands are adjoints of the loss, contract as f16x2.
        amax_out [pad_rows(P) / 32] receives per 32 points the largest value the launch stored, amax_in is such an array of an
        earlier sweep whose outputs enter this one as X2."""
        self.rec.append((6, a, kw))

    # positional parameter names of the recorded calls, and the descriptor field every tensor argument lands in
    _PARAMS = (("x", "L", "in_scale", "tangent", "x_div"), ("G0",), ("A0", "lda0"),
               ("A0", "lda0", "sign", "wrow", "scale", "xscale"), ("epi", "Bp", "K", "N"), ("t",), ("amax_in", "amax_out"))
    _CHAIN_FIELD = ({"x": "x", "tangent": "v"}, {"G0": "G0"}, {"A0": "A0"}, {"A0": "A0", "sign": "seed_sign", "wrow": "seed_wrow"},
                    None, {"t": "absmax_out"}, {"amax_in": "tile_amax_in", "amax_out": "tile_amax_out"})

    def _signature(self):
        """-> (structure, pointers): the record with every tensor replaced by its element count, and the tensors' addresses
        in record order; (None, None) if a tensor is not a contiguous device tensor (the filling path raises the error)."""
        T, PT, plain = torch.Tensor, torch.nn.Parameter, _PLAIN_TYPES
        sig = [self.P, self.init, self.k0, self.tile_rows, PRECISION, FWD_F16X2, BWD_F16X2, HEAD16, STATE16, BLOCKED_STATE, TN_SPLIT]
        ptrs = []
        add, addp = sig.append, ptrs.append
        for kind, a, kw in self.rec:
            add(kind)
            for vals in (a, kw.values()):
                for v in vals:
                    tv = type(v)
                    if tv in plain:
                        add(v)
                    elif tv is T or tv is PT or isinstance(v, T):
                        if not (v.is_cuda and v.is_contiguous()):
                            return None, None
                        addp(v.data_ptr())
                        add(v.numel())
                    else:
                        add(v)
            add(tuple(kw))          # (which keyword arguments, in which order: the pointer slots follow it)
        return sig, ptrs

    def _fill(self):
        """-> the pointer slots of the filled descriptor, in the order _signature lists the tensors: (ctypes object, field,
        field value - tensor address) each, so that a later launch of the same structure only rewrites addresses"""
        self.c = Chain()
        self.c.P, self.c.init, self.c.k0, self.c.tile_rows = self.P, CH_INIT[self.init], self.k0, self.tile_rows
        self.c.x_div = 1
        fns = (self._do_posenc, self._do_init_store, self._do_init_load, self._do_init_seed, self._do_step, self._do_absmax,
               self._do_tile_scale)
        T = torch.Tensor
        slots = []
        for kind, a, kw in self.rec:
            n0 = self.n
            fns[kind](*a, **kw)
            obj = self.c.step[n0] if kind == 4 else self.c
            names = list(self._PARAMS[kind][:len(a)]) + list(kw)
            vals = list(a) + list(kw.values())
            for nm, v in zip(names, vals):
                if isinstance(v, T):
                    field = nm if kind == 4 else self._CHAIN_FIELD[kind][nm]
                    cur = getattr(obj, field)
                    # (a field the filler left NULL for this argument is not a pointer slot)
                    slots.append((obj, field, cur - v.data_ptr()) if cur is not None else None)
        return slots

    def _p(self, t, off=0):
        if t is None:
            return None
        self.keep.append(t)
        return ptr(t) + t.element_size() * off

    def _do_posenc(self, x, L, in_scale, tangent=None, x_div=1):
        """source of the positional encodings: x [P / x_div, 3] (x_div = samples per ray for per-ray directions)."""
        self.c.x, self.c.v = self._p(x), self._p(tangent)
        self.c.pe_L, self.c.pe_jvp, self.c.pe_in_scale = L, (1 if tangent is not None else 0), in_scale
        self.c.x_div = x_div

    def _do_init_store(self, G0):
        if G0 is not None:
            self.c.G0, self.c.ldg0 = self._p(G0), G0.shape[1]
            if _is16(G0) and not _isp4(G0):
                raise _lib.NudfError("a bf16 copy of the initial tile is 4-point packed (mlp.pack16 / _buf)")
            if _is16(G0) or _isblk(G0):
                if self.c.init != CH_INIT["SEED"]:
                    raise _lib.NudfError("a bf16 / blocked copy of the initial tile exists for the SEED initialisation only")
                self.c.init_state16 |= 2 if _is16(G0) else 8

    def _do_absmax(self, t):
        self.c.absmax_out = self._p(t)

    def _do_tile_scale(self, amax_in=None, amax_out=None):
        need = pad_rows(self.P) // 32
        for t in (amax_in, amax_out):
            if t is not None and (t.dtype != torch.float32 or t.numel() < need):
                raise _lib.NudfError("tile_scale: amax_in / amax_out hold one float per 32 points of the row-padded launch")
        self.c.tile_scale = 1
        self.c.tile_amax_in, self.c.tile_amax_out = self._p(amax_in), self._p(amax_out)

    def _do_init_load(self, A0, lda0):
        self.c.A0, self.c.lda0 = self._p(A0), lda0

    def _do_init_seed(self, A0, lda0, sign, wrow, scale, xscale):
        self.c.A0, self.c.lda0 = self._p(A0), lda0
        if _is16(A0):
            if not _isp4(A0):
                raise _lib.NudfError("a bf16 seed operand is 4-point packed (mlp.pack16 / _buf)")
            self.c.init_state16 |= 1
        if _isblk(A0):
            self.c.init_state16 |= 4
        self.c.seed_sign, self.c.seed_wrow = self._p(sign), self._p(wrow)
        self.c.seed_scale, self.c.seed_xscale = scale, xscale

    def _do_step(self, epi, Bp, K, N, bias=None, bias_off=0, X1=None, X2=None, C1=None, C2=None, ldc1=0, ldc2=0, r1_row=None,
             ldr1=1, r1_col=None, iparam=0, act_write=1, act_col0=0, pe_tail_col=-1, pe_tail_scale=0.0, pe_dst=None,
             scale=1.0, xscale=1.0, c1_off=0, c2_off=0, x2_off=0, row_w=None, row_sums=None, X3=None):
        if self.n >= CH_MAX_STEPS:
            raise _lib.NudfError("too many chain steps")
        s = self.c.step[self.n]
        s.Bp, s.bias = self._p(Bp), self._p(bias, bias_off)
        s.X1, s.X2, s.C1, s.C2 = self._p(X1), self._p(X2, x2_off), self._p(C1, c1_off), self._p(C2, c2_off)
        s.pe_dst, s.ld_pe = self._p(pe_dst), (pe_dst.shape[1] if pe_dst is not None else 0)
        s.ldx1 = X1.shape[1] if X1 is not None else 0
        s.ldx2 = X2.shape[1] if X2 is not None else 0
        s.X3, s.ldx3 = self._p(X3), (X3.shape[1] if X3 is not None else 0)       # BWD: second-order term formed in the epilogue
        if X3 is not None and (epi != "BWD" or X1 is None or X2 is None or _isblk(X3)):
            raise _lib.NudfError("X3: the third stored operand of a BWD step (X1 = activations, X2 = R, X3 = DA), row-major")
        s.ldc1 = ldc1 or (C1.shape[1] if (C1 is not None and C1.dim() == 2) else 0)
        s.ldc2 = ldc2 or (C2.shape[1] if (C2 is not None and C2.dim() == 2) else 0)
        s.r1_row, s.ldr1, s.r1_col = self._p(r1_row), ldr1, self._p(r1_col)
        s.row_w, s.row_sums = self._p(row_w), self._p(row_sums)      # SIGMOIDN: compositing sum inside the epilogue
        s.K, s.N, s.epi, s.iparam = K, N, CH[epi], iparam
        s.prec = getattr(Bp, "prec", 0)
        # 16-bit stored state (config-5 mode): X1, X2, C1, the TANGENT mirror C2 and pe_dst of a step are bf16 TOGETHER
        state = [t for t in (X1, X2, X3, C1, C2 if epi == "TANGENT" else None, pe_dst) if t is not None]
        if epi in ("RELU", "MULMASK", "ADDMASK") and (_is16(X1) or _is16(C1)):
            # the ReLU family (colour net): bf16 per array -- X2 (d VIN), the C2 mirror and pe_dst (VIN) stay fp32
            if any(_is16(t) for t in (X2, C2, pe_dst)) or not all(_isp4(t) for t in (X1, C1) if _is16(t)) or s.prec == 0 \
                    or (epi == "RELU" and _is16(X1)):
                raise _lib.NudfError("bf16 stored state of a RELU / MULMASK / ADDMASK step: X1 and / or C1 (4-point packed, "
                                     "mlp.pack16), 16-bit mode only")
            s.layout = (_lib.CH_P4_X1 if _is16(X1) else 0) | (_lib.CH_P4_C1 if _is16(C1) else 0)
        elif any(_is16(t) for t in state):
            if not all(_is16(t) and _isp4(t) for t in state) or epi not in ("SOFTPLUS", "MULSP", "TANGENT", "BWD") or s.prec == 0:
                raise _lib.NudfError("bf16 stored state (4-point packed, mlp.pack16): every state array of a SOFTPLUS / "
                                     "MULSP / TANGENT / BWD step of the 16-bit mode, or none")
            s.layout = _lib.CH_STATE16
        else:
            s.layout = ((1 if _isblk(X1) else 0) | (2 if _isblk(X2) else 0) | (4 if _isblk(C1) else 0) |
                        (8 if (_isblk(C2) and epi == "TANGENT") else 0) | (16 if _isblk(pe_dst) else 0))
            if _isblk(C2) and epi != "TANGENT":
                raise _lib.NudfError("blocked layout: C2 of a TANGENT step only")
        s.act_write, s.act_col0, s.pe_tail_col, s.pe_tail_scale = act_write, act_col0, pe_tail_col, pe_tail_scale
        s.scale, s.xscale = scale, xscale
        self.n += 1
        fl = 2.0 * self.c.P * getattr(Bp, "k_true", K) * getattr(Bp, "n_true", N)
        self.flops += fl
        self.xflops += fl * MFMA_PRODUCTS[int(s.prec)]
        if PROFILE is not None:
            n_true = getattr(Bp, "n_true", N)
            for t in (X1, X2, X3, C1, C2, pe_dst):
                if t is not None:
                    self.nbytes += float(self.c.P) * (min(n_true, t.shape[1]) if t.dim() == 2 else 1) * t.element_size()
            self.epis.append(epi)
            self.blocked = self.blocked or (s.layout & 31) != 0

    def launch(self):
        global chain_memo_hits
        memo_ok = CHAIN_MEMO and self.site is not None and PROFILE is None and CHAIN_DEBUG is None
        sig = ptrs = None
        if memo_ok:
            sig, ptrs = self._signature()
            # a site is called several times per step (the UDF value at 32 768, 3 x 8 192 and 65 536 points; with and without
            # saved state): a few descriptors per (site, size, record length), searched in order
            mkey = (self.site, self.P, len(self.rec))
            ms = _CHAIN_MEMO.get(mkey)
            if sig is not None and ms is not None:
                for m in ms:
                    if m[0] == sig:
                        # same launch, possibly other addresses (small tensors come from the allocator's small pool at
                        # varying addresses): rewrite the pointer fields that moved, launch the memoized descriptor
                        last = m[3]
                        if last != ptrs:
                            for i, (pnew, pold) in enumerate(zip(ptrs, last)):
                                if pnew != pold and m[2][i] is not None:
                                    obj, field, delta = m[2][i]
                                    setattr(obj, field, pnew + delta)
                            m[3] = ptrs
                        chain_memo_hits += 1
                        call("nudf_mlp_chain", m[1])       # (the record holds every operand alive until this builder dies)
                        return
        slots = self._fill()
        self.c.n_steps = self.n
        if CHAIN_DEBUG is not None:
            self.c.dbg = ptr(CHAIN_DEBUG)
        if PROFILE is not None:
            _timed("mlp_chain", self.flops, lambda: call("nudf_mlp_chain", self.c), self._label(), self.nbytes, self.xflops)
        else:
            call("nudf_mlp_chain", self.c)
        if memo_ok and sig is not None and len(slots) == len(ptrs):
            ms = _CHAIN_MEMO.setdefault(mkey, [])
            if len(ms) >= 4:
                ms.pop(0)
            ms.append([sig, self.c, slots, ptrs])
            if len(_CHAIN_MEMO) > 512:          # (sizes that came and went: validation renders, tests)
                _CHAIN_MEMO.clear()
        self.keep = []

    def _label(self):
        """kernel instantiation nudf_mlp_chain dispatches this launch to (mirrors csrc/mlp_chain.hip) + the sweep."""
        P, e = self.c.P, set(self.epis)
        x1 = bool(e & {"MULSP", "TANGENT", "BWD", "MULMASK", "ADDMASK"})
        x2 = bool(e & {"TANGENT", "BWD", "ADDMASK", "RELUADD"})
        pair = int(os.environ.get("NUDF_CHAIN_PAIR", "0"))
        if self.c.tile_rows == 130 or ((self.blocked or self.c.tile_rows == 66) and pair >= 1 and P >= 32768):
            kern = "mlp_chain_pair_kernel<%d>" % (2 if x2 else (1 if x1 else 0))
        elif self.c.tile_rows == 0 and pair >= 2 and P >= 32768 and PRECISION == "fp32":
            kern = "mlp_chain_pair_kernel<%d>" % (2 if x2 else (1 if x1 else 0))
        elif self.blocked or self.c.tile_rows == 66:
            kern = "mlp_chain_tq_kernel<%d>" % (2 if x2 else (1 if x1 else 0))
        elif self.c.tile_rows == 128:
            kern = "mlp_chain_rows_kernel"
        else:
            mode = {"fp32": 0, "mixed16": 1, "bf16x3": 2}[PRECISION]
            t32 = self.c.tile_rows == 32 or (self.c.tile_rows != 64 and P <= 256 * 64)
            roww = any(self.c.step[i].row_w for i in range(self.n))
            if mode == 1 and not t32 and not roww and _chain_t16_now():
                precs = {int(self.c.step[i].prec) for i in range(self.n)}
                if len(precs) == 1 and precs <= {1, 2}:
                    x3 = any(self.c.step[i].X3 for i in range(self.n))
                    mode = 4 if ("TANGENT" in e or x3) else 3      # the 16-bit-tile kernel (one operand type in every step)
            kern = "mlp_chain_kernel<%d, %d>" % (32 if t32 else 64, mode)
        sweep = ("tangent" if ("TANGENT" in e or ("MULSP" in e and self.init == "POSENC")) else "adjoint" if "BWD" in e
                 else "input-gradient" if "MULSP" in e
                 else "relu-backward" if e & {"MULMASK", "ADDMASK"} else "udf-forward" if "SOFTPLUS" in e
                 else "relu-forward")
        return "%s %s P=%d" % (kern, sweep, P)


def _chain_t16_now():
    """the library's CURRENT 16-bit-tile setting (env NUDF_CHAIN_T16 at load, nudf_set_chain_t16 afterwards): the setter
    returns the old value, so set-and-restore reads it (host side only; used by the profiling labels)"""
    L = _lib.lib()
    old = int(L.nudf_set_chain_t16(1))
    L.nudf_set_chain_t16(old)
    return bool(old)


# scratch of the weight-gradient GEMMs' deterministic two-pass reduction (every workgroup's partial tile, ~33 MB at the
# headline size): ONE buffer per device, grown on demand -- all launches of this module go to torch's current stream, so
# consecutive groups never overlap.  TN_DETERMINISTIC = False falls back to fp32 atomics (A-B measurements).
TN_DETERMINISTIC = True
_TN_WS = {}


def _tn_workspace(g, dev):
    need = int(_lib.lib().nudf_gemm_tn_grouped_workspace(C.byref(g)))
    if need <= 0:
        return None, 0
    ws = _TN_WS.get(dev)
    if ws is None or ws.numel() < need:
        ws = _TN_WS[dev] = torch.empty(need, device=dev)
    return ws, ws.numel()


TN_ASSIGN = os.environ.get("NUDF_TN_ASSIGN", "1") == "1"     # A/B + tests: 0 = zero-filled gradient buffers, accumulate


def tn_can_assign():
    """the weight-gradient launches can ASSIGN their outputs (two-pass deterministic reduction; not the atomic A/B path)"""
    return TN_ASSIGN and TN_DETERMINISTIC and not (int(os.environ.get("NUDF_TN_FLAGS", "0")) & (8 | 2))


def gemm_tn_grouped(jobs, M, assign=False, rows_per_block=0, f16x2=False, amax_a=None, amax_b=None):
    """jobs: [(A1 [M, lda], NA, B1 [M, ldb], NB, C [NA_pad, ldc], dbias | None)] -> one launch per <= 12 problems.
    assign: C / dbias are assigned, not accumulated into (every C must then appear in ONE job; see `alloc_grads`).
    rows_per_block: 0 = the library chooses the row chunks (tests pin them to compare kernels bit for bit).
    f16x2 (bf16x3 mode, fp32 row-major operands): three fp16 MFMA products per fp32 product (NudfGemmTNGroup.prec 4); amax_a /
    amax_b = one-element device tensors holding max |x| over all A / all B operands of the call (None: that side's operands
    lie in fp16's range unscaled -- activations)."""
    for base in range(0, len(jobs), _lib.TN_MAX_PROBLEMS):
        chunk = jobs[base:base + _lib.TN_MAX_PROBLEMS]
        g = _lib.GemmTNGroup()
        g.n_problems, g.M, g.rows_per_block = len(chunk), M, rows_per_block
        g.prec = _tn_prec()                            # mixed16: bf16 operands for the weight gradients as well
        if f16x2 and g.prec == 3 and not any(_is16(t) or _isblk(t) for j in chunk for t in (j[0], j[2])):
            g.prec = 4
            g.amax_a, g.amax_b = ptr(amax_a), ptr(amax_b)
        flops = nbytes = 0.0
        for i, (A1, NA, B1, NB, Cm, db) in enumerate(chunk):
            q = g.prob[i]
            q.A1, q.B1, q.C, q.dbias = ptr(A1), ptr(B1), ptr(Cm), ptr(db)
            q.flags = ((1 if _is16(A1) else 0) | (2 if _is16(B1) else 0) | (4 if _isblk(A1) else 0) | (8 if _isblk(B1) else 0) |
                       (16 if _isp4(A1) else 0) | (32 if _isp4(B1) else 0))
            q.lda1, q.ldb1, q.ldc, q.NA, q.NB = A1.shape[1], B1.shape[1], Cm.shape[1], NA, NB
            flops += 2.0 * M * NA * NB
            nbytes += float(M) * (NA * A1.element_size() + NB * B1.element_size())
        if TN_DETERMINISTIC:
            ws, n = _tn_workspace(g, chunk[0][0].device)
            g.workspace, g.workspace_floats = ptr(ws), n
        g.assign = 1 if assign else 0
        if PROFILE is not None:
            _timed("gemm_tn", flops, lambda: call("nudf_gemm_tn_grouped", g),
                   "gemm_tn%s_group_kernel %d problems M=%d (%.1f GFLOP)" % ({3: "3", 4: "2"}.get(int(g.prec), ""), len(chunk), M, flops / 1e9),
                   nbytes, flops * MFMA_PRODUCTS.get(int(g.prec), 1) if (int(g.prec) != 3 or TN_SPLIT) else flops)
        else:
            call("nudf_gemm_tn_grouped", g)


class PackedLinear:
    """one (weight-normed or plain) nn.Linear packed for the GEMM kernels."""

    def __init__(self, lin: torch.nn.Module, perm: Optional[List[int]] = None):
        self.lin = lin
        self.weight_norm = hasattr(lin, "weight_v")
        v = lin.weight_v if self.weight_norm else lin.weight
        self.out, self.inp = v.shape
        self.out_pad, self.in_pad = pad32(self.out), pad32(self.inp)
        self._perm_list = perm
        self.perm = None
        self.W = self.Wt = self.inv_norm = None
        self._ver = None
        self._frags = {}
        self._plist = None

    def params(self):
        # (nn.Module.__getattr__ is a Python function: ~100 look-ups per train step; the module's own parameter dict says
        # whether the cached list still holds the registered objects)
        reg = self.lin._parameters
        pl = self._plist
        if self.weight_norm:
            if pl is None or reg.get("weight_v") is not pl[0] or reg.get("weight_g") is not pl[1] or reg.get("bias") is not pl[2]:
                pl = self._plist = [self.lin.weight_v, self.lin.weight_g, self.lin.bias]
        elif pl is None or reg.get("weight") is not pl[0] or reg.get("bias") is not pl[1]:
            pl = self._plist = [self.lin.weight, self.lin.bias]
        return list(pl)

    def _ensure_buffers(self, dev):
        if self.W is None or self.W.device != dev:
            self.W = torch.zeros(self.out_pad, self.in_pad, device=dev)
            self.Wt = torch.zeros(self.in_pad, self.out_pad, device=dev)
            self.inv_norm = torch.empty(self.out, device=dev)
            self.Wt.k_true, self.W.k_true = self.inp, self.out      # unpadded reduction lengths (flop accounting)
            self._frags = {}
            if self._perm_list is not None:
                self.perm = torch.tensor(self._perm_list, dtype=torch.int32, device=dev)

    def pack(self):
        """(re)pack if the parameters changed; returns self."""
        ps = self.params()
        v = ps[0]
        ver = tuple((p.data_ptr(), p._version) for p in ps[:-1])
        if ver == self._ver and self.W is not None and self.W.device == v.device:
            return self
        dev = v.device
        if self.W is None or self.W.device != dev:
            self.W = torch.zeros(self.out_pad, self.in_pad, device=dev)
            self.Wt = torch.zeros(self.in_pad, self.out_pad, device=dev)
            self.inv_norm = torch.empty(self.out, device=dev)
            self.Wt.k_true, self.W.k_true = self.inp, self.out      # unpadded reduction lengths (flop accounting)
            if self._perm_list is not None:
                self.perm = torch.tensor(self._perm_list, dtype=torch.int32, device=dev)
        g = ps[1] if self.weight_norm else None
        call("nudf_weightnorm_pack", ptr(v.detach().contiguous()), ptr(g.detach().contiguous()) if g is not None else None,
             self.out, self.inp, ptr(self.perm), ptr(self.W), self.in_pad, ptr(self.Wt), self.out_pad,
             ptr(self.inv_norm))
        self._ver = ver
        self._frags = {}
        return self

    def invalidate(self):
        """forget the packed weights.  The cache is keyed on the parameters' (data_ptr, _version); in-place writes
        through `p.data` (p.data.copy_(), EMA, manual re-initialisation) do NOT bump `_version`, so whoever does that
        must call this (or `module.invalidate()` / `engine.invalidate()`)."""
        self._ver = None
        self._frags = {}

    def frag(self, kind):
        """fragment-ordered copies for the fused chains (cached until the parameters change):
        'fwd'  B = W^T [inp, out]            'bwd'  B = W [out, inp]
        'fwd_head0' / 'fwd_feat'  column 0 / columns 1.. of W^T (the UDF head, fields.py:184-190)
        'bwd_feat'  rows 1.. of W"""
        f = self._frags.get(kind)
        if f is None:
            if "@" in kind:
                raise _lib.NudfError("16-bit fragment %r was not packed (pack_group packs them)" % kind)
            if kind == "fwd":
                f = pack_frag(self.Wt, self.out_pad, self.inp, self.out)
            elif kind == "bwd":
                f = pack_frag(self.W, self.in_pad, self.out, self.inp)
            elif kind == "fwd_head0":
                f = pack_frag(self.Wt, self.out_pad, self.inp, 1)
            elif kind == "fwd_feat":
                f = pack_frag(self.Wt, self.out_pad, self.inp, self.out - 1, off=1)
            elif kind == "bwd_feat":
                f = pack_frag(self.W, self.in_pad, self.out - 1, self.inp, off=self.in_pad)
            elif kind.startswith("bwd_hid:"):
                f = pack_frag(self.W, self.in_pad, self.out, int(kind.split(":")[1]))
            elif kind.startswith("fwd_in:"):
                _, i0, K = kind.split(":")
                f = pack_frag(self.Wt, self.out_pad, int(K), self.out, off=int(i0) * self.out_pad)
            else:
                raise KeyError(kind)
            self._frags[kind] = f
        return f

    @property
    def bias(self):
        return self.lin.bias.detach()

    def new_grad_buffers(self):
        dev = self.W.device
        return torch.zeros(self.out_pad, self.in_pad, device=dev), torch.zeros(self.out, device=dev)

    def unpack_grads(self, dW, db):
        """packed dW/db -> gradients in params() order."""
        ps = self.params()
        v = ps[0].detach().contiguous()
        dv = torch.empty_like(v)
        if self.weight_norm:
            g = ps[1].detach().contiguous()
            dg = torch.empty_like(g)
            call("nudf_weightnorm_unpack_grad", ptr(dW), self.in_pad, ptr(v), ptr(g), ptr(self.inv_norm), self.out,
                 self.inp, ptr(self.perm), ptr(dv), ptr(dg))
            return [dv, dg, db]
        call("nudf_weightnorm_unpack_grad", ptr(dW), self.in_pad, ptr(v), None, None, self.out, self.inp,
             ptr(self.perm), ptr(dv), None)
        return [dv, db]


# MFMA operand precision of the fused chains.  "fp32" is the exact fp32 MFMA path (v_mfma_f32_32x32x2_f32).  "mixed16" is
# BASELINE config 5 (16-bit MLP weights on the CDNA4 matrix cores): fp16 operands in the forward sweeps (value,
# input gradient, colour), bf16 operands in the backward sweeps (tiny adjoints need fp32's exponent range), fp32
# accumulation, fp32 activations / stored state / epilogues / weight gradients / optimizer.  The abs-head column
# (the UDF value itself) stays in fp32.
# "bf16x3": fp32 EMULATED on the bf16 matrix pipe (NudfChainStep.prec = 3): weights and activations split exactly into
# three bf16 parts, six partial products per fp32 product, fp32 accumulation -- fp32-level accuracy (the dropped terms are
# the size of one fp32 rounding) at 12 instead of 32 matrix-pipe cycles per k; state, epilogues, optimizer: fp32.
# This is the library DEFAULT (and bench.py's headline): its results are fp32-accurate (tests/test_gpu_bf16x3.py: errors against
# float64 equal the exact kernels') but NOT bit-comparable with the "fp32" kernels -- tests that assert bit-identity between
# kernels pin set_precision("fp32").  Non-finite inputs: the split of +-inf is inf - inf = NaN in the remainder parts, so an
# infinite activation or weight comes out as NaN where the fp32 kernels propagate inf; both are non-finite, and the status
# word (include/nudf.h: nudf_set_status_flag) reports either.  NUDF_PRECISION=fp32 selects the exact kernels.
PRECISION = os.environ.get("NUDF_PRECISION", "bf16x3")
_PREC = {"f32": 0, "f16": 1, "bf16": 2, "bf16x3": 3, "f16x2": 4}
# bf16x3 mode, forward-order sweeps (UDF value, input gradient, colour / NeRF forward: encodings, softplus / ReLU activations,
# seed-scaled weight rows and weight-normed weights -- all inside fp16's range): the fp32 product on THREE fp16 MFMA products
# instead of six bf16 ones (NudfChainStep.prec = 4: x = hi + 2^-11 lo, the correction terms in their own accumulator).
# The backward sweeps (tangent, adjoint, ReLU backward: 1e-6 ... 1e-9 adjoints, below fp16's normal range) and the
# weight-gradient GEMMs stay on bf16x3.  Accuracy against float64 equals the exact fp32 kernels' (tests/test_gpu_bf16x3.py,
# scripts/numerics/f16x2_emulation.py).  NUDF_FWD_F16X2=0 keeps bf16x3 everywhere (A/B); "grad" additionally keeps the
# input-gradient reverse sweep on bf16x3.
FWD_F16X2 = os.environ.get("NUDF_FWD_F16X2", "1")
# The backward sweeps (tangent, adjoint, ReLU backward) as f16x2 as well: their operands are adjoints of the LOSS (1e-5 ... 1e-12,
# far below fp16's range), but each of these sweeps maps a point's seeds LINEARLY to that point's outputs, so the kernel runs
# every tile of points multiplied by its own power of two (NudfChain.tile_scale, ChainBuilder.tile_scale) and memory holds what
# it held before.  NUDF_BWD_F16X2=0 keeps them on bf16x3 (A/B).
BWD_F16X2 = os.environ.get("NUDF_BWD_F16X2", "1") != "0"
# the adjoint sweep forms the second-order term from R and DA instead of reading an EX array the tangent sweep stored
# (NudfChainStep.X3; UDFEngine._backward_chain): 0 = the stored form (A/B)
EX_FLY = os.environ.get("NUDF_EX_FLY", "1") != "0"
# bf16x3 mode: the weight-gradient GEMMs of the UDF and colour networks on THREE fp16 products (NudfGemmTNGroup.prec 4,
# gemm_tn2_group_kernel) -- the side of every problem that holds loss adjoints is scaled by a power of two taken from the
# maximum the producing sweeps report (NudfChain.absmax_out), the activation side is used as it is.  0 = bf16x3 GEMMs (A/B).
TN_F16X2 = os.environ.get("NUDF_TN_F16X2", "1") != "0"
TN_SPLIT = os.environ.get("NUDF_TN_SPLIT", "1") != "0"      # bf16x3 mode: the weight-gradient GEMMs take split operands too


def _tn_prec():
    if PRECISION == "fp32":
        return 0
    if PRECISION == "bf16x3":
        return 3 if TN_SPLIT else 0
    return 2


# 16-bit mode: the UDF engine's saved-for-backward arrays (X, DA, R, EX, ABAR) are stored as bf16 -- half the HBM
# traffic of the sweeps and of the weight-gradient GEMMs that read them.  STATE16 = False keeps them fp32 (A-B).
STATE16 = os.environ.get("NUDF_STATE16", "1") != "0"
# 16-bit mode: the abs-head column (K = 256, N = 1) contracts in fp16 like every other step of the forward sweep, so that the
# whole sweep has ONE operand type and runs on the 16-bit-TILE chain kernel (mlp_chain_kernel<64, 3>: the LDS activation tile IS
# the fp16 operand, three workgroups per CU).  The activations the head sees are the fp16-rounded ones every hidden layer sees;
# only the head's 256 weights lose their fp32 mantissa.  HEAD16 = False keeps the head on the fp32 MFMA (and the sweep on the
# fp32-tile kernel): A/B, and the reference point of tests/test_gpu_mixed16.py.
HEAD16 = os.environ.get("NUDF_HEAD16", "1") != "0"


def _head_kind():
    return "fwd_head0@f16" if (PRECISION == "mixed16" and HEAD16) else "fwd_head0"


# fp32 mode, large launches: the UDF engine's saved state in the BLOCKED layout + the transposed-product chain kernel
# (1 KB of contiguous memory per wave instruction in the epilogues, 16 KB contiguous operand tiles in the weight-gradient
# GEMM).  NUDF_BLOCKED_STATE=0 keeps row-major buffers and the default kernel (A-B).
BLOCKED_STATE = os.environ.get("NUDF_BLOCKED_STATE", "1") != "0"


def _state_blocked(P):
    # (bf16x3: measured with the split K loop ported into the transposed-product kernel -- chains 2.752 vs 2.758 ms, and the
    # weight-gradient GEMM loses its split-image kernel on blocked operands, 1.14 -> 2.09 ms: row-major state there)
    return BLOCKED_STATE and PRECISION == "fp32" and USE_CHAIN and CHAIN_TILE in (0, 66, 130) and P > 256 * 64


def _state_dtype():
    return torch.bfloat16 if (PRECISION == "mixed16" and STATE16) else torch.float32


def set_precision(name):
    global PRECISION
    if name not in ("fp32", "mixed16", "bf16x3"):
        raise ValueError("precision must be 'fp32', 'bf16x3' or 'mixed16'")
    PRECISION = name


def set_fwd_split(name):
    """the forward-order sweeps of the bf16x3 mode: "1" (default) = f16x2 on value / colour / NeRF forward AND the
    input-gradient sweep, "grad" = f16x2 on the forward sweeps only, "0" = bf16x3 everywhere.  -> the old setting."""
    global FWD_F16X2
    if str(name) not in ("0", "1", "grad"):
        raise ValueError("forward split must be '0', '1' or 'grad'")
    old, FWD_F16X2 = FWD_F16X2, str(name)
    return old


def _sweep_dtype(sweep):
    """operand dtype of a sweep: 'fwd' (value / colour / NeRF forward), 'grad' (the input-gradient reverse sweep of the
    forward pass: adjoints of the UDF VALUE, O(weights) in size) or 'bwd' (tangent / adjoint / ReLU backward of the loss)."""
    if PRECISION == "fp32":
        return "f32"
    if PRECISION == "bf16x3":
        if FWD_F16X2 != "0" and (sweep == "fwd" or (sweep == "grad" and FWD_F16X2 != "grad")):
            return "f16x2"
        if sweep == "bwd" and BWD_F16X2 and FWD_F16X2 != "0":
            return "f16x2"
        return "bf16x3"
    return "f16" if sweep in ("fwd", "grad") else "bf16"


def _kind(base, sweep):
    dt = _sweep_dtype(sweep)
    return base if dt == "f32" else base + "@" + dt


def _frag_spec(pl, kind):
    """(transpose, o0, i0, K, N, dtype) of a fragment-ordered operand cut from the packed [out, in] matrix."""
    dtype = 0
    if "@" in kind:
        kind, dt = kind.split("@")
        dtype = _PREC[dt]
    return _frag_spec32(pl, kind) + (dtype,)


def _frag_spec32(pl, kind):
    if kind == "fwd":
        return (1, 0, 0, pl.inp, pl.out)
    if kind == "bwd":
        return (0, 0, 0, pl.out, pl.inp)
    if kind == "fwd_head0":
        return (1, 0, 0, pl.inp, 1)
    if kind == "fwd_feat":
        return (1, 1, 0, pl.inp, pl.out - 1)
    if kind == "bwd_feat":
        return (0, 1, 0, pl.out - 1, pl.inp)
    if kind.startswith("bwd_hid:"):
        return (0, 0, 0, pl.out, int(kind.split(":")[1]))
    if kind.startswith("fwd_in:"):       # W^T restricted to the (packed-order) input columns [i0, i0 + K)
        _, i0, K = kind.split(":")
        return (1, 0, int(i0), int(K), pl.out)
    raise KeyError(kind)


def _mode_key():
    """everything that decides WHICH fragment copies a network's sweeps read"""
    return (PRECISION, FWD_F16X2, BWD_F16X2, HEAD16)


def pack_group(layers, kinds=None):
    """(re)pack every layer of a network in ONE launch if any parameter changed: weight_norm, W / W^T and the
    fragment-ordered copies named in `kinds` (one tuple of kinds per layer)."""
    # repeat call with nothing changed (five of the six calls of a train step): the previous verdict is replayed from a memo
    # kept on the first layer -- per layer one identity test of its packed version and (data_ptr, _version) of its two
    # weight parameters, ~5 us per network instead of ~100 us of module attribute look-ups (eager host cost)
    memo_key = (len(layers), tuple(kinds) if kinds is not None else None)
    memos = getattr(layers[0], "_group_memo", None) if _lib.HOST_FAST else None
    m = memos.get(memo_key) if memos is not None else None
    if m is not None:
        fresh = True
        for pl, ver_obj, p0, v0, d0, p1, v1, d1 in m:
            if (pl._ver is not ver_obj or p0._version != v0 or p0.data_ptr() != d0
                    or (p1 is not None and (p1._version != v1 or p1.data_ptr() != d1))):
                fresh = False
                break
        if fresh:
            return
    kinds = kinds or [()] * len(layers)
    dev = layers[0].params()[0].device
    stale = False
    kinds = [tuple(dict.fromkeys(ks)) for ks in kinds]     # (bf16x3: the forward and backward sweeps share one dtype)
    for pl, ks in zip(layers, kinds):
        ps = pl.params()
        ver = tuple((p.data_ptr(), p._version) for p in ps[:-1])
        if ver != pl._ver or pl.W is None or pl.W.device != dev or any(k not in pl._frags for k in ks):
            stale = True
        pl._new_ver = ver
    if not stale:
        _remember_group(layers, memo_key)
        return
    # a repack of the SAME buffers (the parameters changed in place: every step after the optimizer's): the filled launch
    # descriptors of the last repack are reused when every address in them still holds -- parameters, packed matrices and
    # fragment copies are all persistent buffers
    psig = _pack_ptr_signature(layers, kinds, dev)
    if m is not None and psig is not None and getattr(layers[0], "_pack_descs", {}).get(memo_key, (None,))[0] == psig:
        for a in layers[0]._pack_descs[memo_key][1]:
            call("nudf_weightnorm_pack_multi", a)
        for pl in layers:
            pl._ver = pl._new_ver
        _remember_group(layers, memo_key)
        return
    descs = []
    for base in range(0, len(layers), _lib.PACK_MAX_LAYERS):
        chunk = layers[base:base + _lib.PACK_MAX_LAYERS]
        a = _lib.PackMulti()
        rows = 0
        keep = []
        for li, pl in enumerate(chunk):
            ks = kinds[base + li]
            if len(ks) > _lib.PACK_MAX_FRAGS:
                raise _lib.NudfError("too many fragment copies for one layer")
            pl._ensure_buffers(dev)
            ps = pl.params()
            v = ps[0].detach().contiguous()
            g = ps[1].detach().contiguous() if pl.weight_norm else None
            keep += [v, g]
            L = a.layer[li]
            L.v, L.g, L.perm = ptr(v), ptr(g), ptr(pl.perm)
            L.W, L.Wt, L.inv_norm = ptr(pl.W), ptr(pl.Wt), ptr(pl.inv_norm)
            L.out, L.in_, L.ldw, L.ldwt = pl.out, pl.inp, pl.in_pad, pl.out_pad
            L.nfrag, L.row_start = len(ks), rows
            for fi, kind in enumerate(ks):
                tr, o0, i0, K, N, dtype = _frag_spec(pl, kind)
                f = pl._frags.get(kind)
                if f is None:
                    nfl = k8(K) // 8 * ((N + 31) // 32) * 256          # fp32 fragments; 16-bit ones take half
                    f = torch.zeros({0: nfl, 3: 3 * (nfl // 2), 4: nfl}.get(dtype, nfl // 2), device=dev, dtype=torch.float32)
                    f.k_true, f.n_true, f.prec = K, N, dtype
                    pl._frags[kind] = f
                F = L.frag[fi]
                F.dst, F.transpose, F.o0, F.i0, F.K, F.N, F.dtype = ptr(f), tr, o0, i0, K, N, dtype
            rows += pl.out
        a.n_layers, a.total_rows = len(chunk), rows
        call("nudf_weightnorm_pack_multi", a)
        descs.append(a)
        del keep
    for pl in layers:
        pl._ver = pl._new_ver
    _remember_group(layers, memo_key)
    psig = _pack_ptr_signature(layers, kinds, dev)
    if psig is not None:
        layers[0].__dict__.setdefault("_pack_descs", {})[memo_key] = (psig, descs)


def _pack_ptr_signature(layers, kinds, dev):
    """every address a filled NudfPackMulti of this network holds (None: some buffer does not exist yet, or a parameter is
    not contiguous and would be packed from a temporary copy)"""
    sig = []
    for pl, ks in zip(layers, kinds):
        if pl.W is None or pl.W.device != dev:
            return None
        for p in pl.params()[:-1]:
            if not p.is_contiguous():
                return None
            sig.append(p.data_ptr())
        sig += [pl.W.data_ptr(), pl.Wt.data_ptr(), pl.inv_norm.data_ptr(), pl.perm.data_ptr() if pl.perm is not None else 0]
        for k in ks:
            f = pl._frags.get(k)
            if f is None:
                return None
            sig.append(f.data_ptr())
    return sig


def _remember_group(layers, memo_key):
    """memo of a network found (or made) fresh by pack_group: per layer the packed-version OBJECT (mark_stale / invalidate /
    a repack replace it) and the identity, _version and address of the weight parameters it was packed from"""
    m = []
    for pl in layers:
        ps = pl.params()[:-1]
        p0 = ps[0]
        p1 = ps[1] if len(ps) > 1 else None
        m.append((pl, pl._ver, p0, p0._version, p0.data_ptr(), p1, p1._version if p1 is not None else 0,
                  p1.data_ptr() if p1 is not None else 0))
    memos = getattr(layers[0], "_group_memo", None)
    if memos is None or len(memos) > 8:
        memos = layers[0]._group_memo = {}
    memos[memo_key] = m


def claim_grad_slot(engine, layers):
    """The engine's segment of the data-parallel gradient bucket (dist.GradBucket) if this backward may write into it,
    else None (fresh gradient tensors, which autograd then ADDS to whatever is there).  The segment is persistent and
    autograd installs views of it as `p.grad`, so it can be written at most once between two resets of the gradients:
      * a second backward of the same engine inside ONE autograd pass (two evaluations of a network in one graph) finds
        `_slot_inflight` set -- autograd has not installed the first result yet, it would sum the aliased views;
      * a backward while some parameter's `.grad` still lives in the segment (gradient accumulation,
        zero_grad(set_to_none=False)) would overwrite that gradient and AccumulateGrad would add the tensor to itself.
    Trainer.step (one evaluation per network, zero_grad(set_to_none=True)) always gets the segment."""
    slot = getattr(engine, "grad_slot", None)
    if slot is None or getattr(engine, "_slot_inflight", False):
        return None
    lo, hi = slot.data_ptr(), slot.data_ptr() + 4 * slot.numel()
    for pl in layers:
        for p in pl.params():
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                return None
    try:     # cleared when the running autograd pass ends; outside a backward pass nothing accumulates
        torch.autograd.Variable._execution_engine.queue_callback(lambda: setattr(engine, "_slot_inflight", False))
        engine._slot_inflight = True
    except RuntimeError:
        pass
    return slot


def unpack_group(layers, grads, slot=None):
    """packed (dW, db) per layer -> parameter gradients in params() order, ONE launch per <= 16 layers.
    `slot`: optional flat fp32 tensor of exactly sum(p.numel() for the layers' params) elements (a segment of the
    data-parallel gradient bucket, dist.GradBucket): the kernel then writes dv / dg / db straight into it and the
    returned gradients are views of it, so the all-reduce needs no `cat` and no copy back."""
    out_per_layer = []
    off = 0
    if slot is not None:
        need = sum(p.numel() for pl in layers for p in pl.params())
        if slot.numel() != need or slot.dtype != torch.float32 or not slot.is_contiguous():
            raise _lib.NudfError("gradient slot has %d elements, the layers need %d" % (slot.numel(), need))

    def take(shape_like):
        nonlocal off
        if slot is None:
            return torch.empty_like(shape_like)
        n = shape_like.numel()
        v = slot[off:off + n].view(shape_like.shape)
        off += n
        return v
    for base in range(0, len(layers), _lib.PACK_MAX_LAYERS):
        chunk = layers[base:base + _lib.PACK_MAX_LAYERS]
        a = _lib.UnpackMulti()
        rows = 0
        keep = []
        for li, pl in enumerate(chunk):
            dW, db = grads[base + li]
            ps = pl.params()
            v = ps[0].detach().contiguous()
            dv = take(v)
            L = a.layer[li]
            L.dW, L.v, L.perm, L.dv = ptr(dW), ptr(v), ptr(pl.perm), ptr(dv)
            L.out, L.in_, L.ldw, L.row_start = pl.out, pl.inp, pl.in_pad, rows
            dg = None
            if pl.weight_norm:
                g = ps[1].detach().contiguous()
                dg = take(g)
                L.g, L.inv_norm, L.dg = ptr(g), ptr(pl.inv_norm), ptr(dg)
                keep += [v, g]
            else:
                keep += [v]
            dbo = take(ps[-1].detach())         # fresh tensor (or bucket view): autograd takes it without a clone
            L.db_in, L.db_out = ptr(db), ptr(dbo)
            keep.append(db)
            out_per_layer.append([dv, dg, dbo] if pl.weight_norm else [dv, dbo])
            rows += pl.out
        a.n_layers, a.total_rows = len(chunk), rows
        call("nudf_weightnorm_unpack_grad_multi", a)
        del keep
    return [t for lay in out_per_layer for t in lay]



# =========================================================================================
# UDF network
# =========================================================================================
class UDFEngine:
    def __init__(self, net):
        self.net = net
        self.L = net.num_layers - 2                     # index of the last linear layer
        self.E = net.embed_dim
        self.layers = [PackedLinear(getattr(net, f"lin{l}")) for l in range(self.L + 1)]
        self.skip = set(net.skip_in)
        self.inv_sqrt2 = 1.0 / math.sqrt(2.0)
        # udf_out (fields.py:184-190): the head kernels store f(h0) / scale and the multiplier f'(h0) ("sign" below: sign h0,
        # 2 h0 or 1) that every sweep behind the head uses
        self.head_type = {"abs": 0, "square": 1, "sdf": 2}[getattr(net, "udf_type", "abs")]

    def params(self):
        out = []
        for pl in self.layers:
            out += pl.params()
        return out

    def invalidate(self):
        for pl in self.layers:
            pl.invalidate()

    def mark_stale(self):
        """re-pack the weights at the next use, keeping the buffers (train.GraphedStep: the pack launch must be part of
        the captured step even when a forward-only render packed the current weights just before the capture)."""
        for pl in self.layers:
            pl._ver = None

    def _embed(self, x, P, X, tangent=None):
        net = self.net
        # layer-0 input, plus the tail of every skip layer's input (cat([x, inputs])/sqrt(2), fields.py:202-203)
        skip_dst = [X[s] for s in sorted(self.skip)]
        d2 = skip_dst[0] if skip_dst else None
        off2 = (self.layers[sorted(self.skip)[0]].inp - self.E) if skip_dst else 0
        call("nudf_posenc", ptr(x), 3, 1, ptr(tangent), net.d_in, net.multires, float(net.scale), P,
             ptr(X[0]), X[0].shape[1], 1.0,
             (ptr(d2) + 4 * off2) if d2 is not None else None, d2.shape[1] if d2 is not None else 0, self.inv_sqrt2)
        for s, dst in zip(sorted(self.skip)[1:], skip_dst[1:]):
            off = self.layers[s].inp - self.E
            call("nudf_posenc", ptr(x), 3, 1, ptr(tangent), net.d_in, net.multires, float(net.scale), P,
                 ptr(dst) + 4 * off, dst.shape[1], self.inv_sqrt2, None, 0, 0.0)

    # -- dispatch: fused LDS-resident chains (default) or per-layer GEMM launches -------------------
    def _chain_ok(self):
        net = self.net
        widths_ok = all(pl.out <= 256 and pl.inp <= 256 for pl in self.layers[:-1]) and self.layers[-1].inp <= 256 \
            and self.layers[-1].out - 1 <= 256
        skip_ok = len(self.skip) <= 1 and all(0 < s < self.L for s in self.skip)
        return USE_CHAIN and net.multires > 0 and net.d_in == 3 and widths_ok and skip_ok and self.L + 2 <= CH_MAX_STEPS

    def forward(self, x, need_grad_state, feat_ld=0, udf_only=False, feat_buf=None):
        if self._chain_ok():
            return self._forward_chain(x, need_grad_state, feat_ld, udf_only, feat_buf)
        return self._forward_layers(x, need_grad_state, feat_ld, udf_only)

    def gradient(self, x, st):
        if self._chain_ok():
            return self._gradient_chain(x, st)
        return self._gradient_layers(x, st)

    def backward(self, x, st, DA, d_udf, d_feat, d_feat_ld, d_g):
        if self._chain_ok():
            return self._backward_chain(x, st, DA, d_udf, d_feat, d_feat_ld, d_g)
        return self._backward_layers(x, st, DA, d_udf, d_feat, d_feat_ld, d_g)

    def _frag_kinds(self):
        """fragment-ordered weight copies each layer needs for the four sweeps (cached per operand mode)."""
        kc = self.__dict__.setdefault("_kinds_cache", {})
        k = kc.get(_mode_key())
        if k is None:
            k = kc[_mode_key()] = tuple(tuple(dict.fromkeys(ks)) for ks in self._frag_kinds_build())
        return k

    def _frag_kinds_build(self):
        kinds = []
        for l, pl in enumerate(self.layers):
            if l == self.L:      # the abs-head column (udf itself): fp32, or fp16 in the 16-bit mode (HEAD16)
                ks = [_head_kind(), _kind("fwd_feat", "fwd"), _kind("bwd_feat", "bwd")]
            elif l in self.skip:
                ks = [_kind("fwd", "fwd"), _kind("bwd", "grad"), _kind("bwd_hid:%d" % self.layers[l - 1].out, "bwd")]
                if PRECISION != "fp32":
                    ks.append(_kind("fwd", "bwd"))
            else:
                # forward value + tangent sweep use W^T ("fwd"), input-gradient + adjoint sweeps use W ("bwd")
                ks = [_kind("fwd", "fwd"), _kind("bwd", "grad")]
                if PRECISION != "fp32":
                    ks += [_kind("fwd", "bwd"), _kind("bwd", "bwd")]
            kinds.append(tuple(ks))
        return kinds

    def _skip_col(self, l):
        """tile column where PE(x)/sqrt(2) starts in the input of skip layer l."""
        return self.layers[l].inp - self.E

    def _forward_chain(self, x, need_grad_state, feat_ld=0, udf_only=False, feat_buf=None):
        """one launch: posenc -> 8 softplus layers -> abs head, activations resident in LDS.
        feat_buf: [pad_rows(P), max(feat_ld, F)] with [x | 0] already in columns F.. (nudf_merge_points), or None."""
        P, dev, L = x.shape[0], x.device, self.L
        net = self.net
        pack_group(self.layers, self._frag_kinds())
        sd = _state_dtype()     # X[0] (the encoding, written by the tile initialisation) stays fp32, row-major
        blk = _state_blocked(P)
        X = ([_buf(P, self.layers[0].inp, dev, zero=False)] +
             [_buf(P, pl.inp, dev, zero=False, dtype=sd, blocked=blk) for pl in self.layers[1:]]) if need_grad_state else None
        cb = ChainBuilder(P, "POSENC", k8(self.E), site=("udf_fwd", _memo_token(self)))
        cb.posenc(x, net.multires, float(net.scale))
        if need_grad_state:
            cb.init_store(X[0])
        for l in range(L):
            pl = self.layers[l]
            nxt_skip = (l + 1) in self.skip
            cb.step("SOFTPLUS", pl.frag(_kind("fwd", "fwd")), k8(pl.inp), pl.out, bias=pl.bias,
                    C1=X[l + 1] if need_grad_state else None,
                    scale=self.inv_sqrt2 if nxt_skip else 1.0,
                    pe_tail_col=self._skip_col(l + 1) if nxt_skip else -1, pe_tail_scale=self.inv_sqrt2,
                    pe_dst=X[l + 1] if (need_grad_state and nxt_skip) else None)
        pl = self.layers[L]
        Pp = pad_rows(P)
        udf = torch.empty(Pp, device=dev)
        sign = torch.empty(Pp, device=dev) if need_grad_state else None
        F = pl.out - 1
        feat = None
        if not udf_only:
            ld = max(feat_ld, F)
            if feat_buf is not None and tuple(feat_buf.shape) == (Pp, ld) and ld >= F + 3 and feat_buf.is_contiguous():
                feat = feat_buf
            else:
                feat = torch.empty((Pp, ld), device=dev)
                if ld >= F + 3:
                    call("nudf_copy_cols", ptr(x), 3, 1, ptr(feat) + 4 * F, ld, 3, P, 1.0)
                    _zero_cols(feat, F + 3)
                else:
                    _zero_cols(feat, F)
            cb.step("NONE", pl.frag(_kind("fwd_feat", "fwd")), k8(pl.inp), F, bias=pl.bias, bias_off=1, C1=feat,
                    act_write=0)
        cb.step("UDFHEAD", pl.frag(_head_kind()), k8(pl.inp), 1, bias=pl.bias, C1=sign, C2=udf, ldc1=1, ldc2=1,
                act_write=0, scale=1.0 / float(net.scale), iparam=self.head_type)
        cb.launch()
        return dict(udf=udf[:P], sign=(sign[:P] if sign is not None else None),
                    feat=(feat[:P] if feat is not None else None), X=X, P=P)

    def _gradient_chain(self, x, st):
        """one launch: seed -> reverse sweep DA[L-1..0] -> d/d(embedding); then the encoding's VJP."""
        P, L, dev = st["P"], self.L, x.device
        X = st["X"]
        net = self.net
        DA = [_buf(P, self.layers[l].out, dev, zero=False, dtype=X[L].dtype, blocked=_isblk(X[L])) for l in range(L)]
        plL = self.layers[L]
        Epad = pad32(self.E)
        cb = ChainBuilder(P, "SEED", k8(self.layers[L - 1].out), site=("udf_grad", _memo_token(self)))
        cb.init_seed(X[L], X[L].shape[1], st["sign"], plL.W, 1.0 / float(net.scale), self._xs(L - 1))
        cb.init_store(DA[L - 1])
        demb_skip = None
        for l in range(L - 1, 0, -1):
            pl = self.layers[l]
            if l in self.skip:
                demb_skip = torch.empty(pad_rows(P), Epad, device=dev)
                cb.step("MULSP", pl.frag(_kind("bwd", "grad")), k8(pl.out), pl.inp, X1=X[l], C1=DA[l - 1], C2=demb_skip,
                        iparam=self.layers[l - 1].out, scale=self.inv_sqrt2, xscale=self._xs(l - 1))
            else:
                cb.step("MULSP", pl.frag(_kind("bwd", "grad")), k8(pl.out), pl.inp, X1=X[l], C1=DA[l - 1],
                        xscale=self._xs(l - 1))
        demb0 = torch.empty(pad_rows(P), Epad, device=dev)
        pl0 = self.layers[0]
        cb.step("NONE", pl0.frag(_kind("bwd", "grad")), k8(pl0.out), pl0.inp, C1=demb0, act_write=0)
        cb.launch()
        g = torch.empty(P, 3, device=dev)
        call("nudf_posenc_vjp", ptr(x), 3, net.d_in, net.multires, float(net.scale), P,
             ptr(demb0), Epad, 1.0, ptr(demb_skip), Epad, 1.0, ptr(g))
        return g, DA

    def _backward_chain(self, x, st, DA, d_udf, d_feat, d_feat_ld, d_g):
        """tangent sweep (second order) and adjoint sweep as one launch each; weight gradients as TN GEMMs."""
        P, L, dev = st["P"], self.L, x.device
        X, sign = st["X"], st["sign"]
        layers = self.layers
        net = self.net
        grouped = os.environ.get("NUDF_UDF_TN_GROUPED", "1") == "1"      # A/B switches (profiling)
        head4_path = grouped and os.environ.get("NUDF_UDF_HEAD4", "1") == "1"
        assign = head4_path and tn_can_assign()       # the first grouped launch writes every gradient element: no fill
        grads = alloc_grads(layers, zero=not assign)
        second = d_g is not None and DA is not None
        R = EX = None
        # EX_FLY: the second-order term EX[l] = (R[l] W^T) * DA[l] * softplus'' / softplus' is not stored by the tangent sweep
        # and re-read by the adjoint sweep -- the adjoint sweep forms it from R[l + 1] and DA[l], arrays that exist anyway
        # (NudfChainStep.X3).  The tangent sweep is then MULSP steps: one array in (X), one out (R) per layer instead of
        # two and two.  Split / 16-bit modes on the workgroup-shared kernel (the fp32 kernels keep the stored form).
        ex_fly = second and EX_FLY and PRECISION != "fp32" and not _isblk(X[L]) and CHAIN_TILE in (0, 32, 64)
        # f16x2 weight-gradient GEMMs: [max |adjoint-sweep arrays|, max |tangent-sweep arrays|], raised by the two sweeps
        tn2 = (TN_F16X2 and PRECISION == "bf16x3" and TN_SPLIT and grouped and head4_path and not _isblk(X[L]) and not _is16(X[L])
               and CHAIN_TILE in (0, 32, 64))
        amax = torch.zeros(2, device=dev) if tn2 else None
        # f16x2 backward sweeps: per-tile scaling; the tangent sweep's per-tile maxima bound what its R arrays add to the adjoint
        bsc = _sweep_dtype("bwd") == "f16x2"
        TA = torch.empty(pad_rows(P) // 32, device=dev) if (bsc and second) else None
        if second:
            sd, blk = X[L].dtype, _isblk(X[L])
            R = ([_buf(P, layers[0].inp, dev, zero=False)] +
                 [_buf(P, pl.inp, dev, zero=False, dtype=sd, blocked=blk) for pl in layers[1:]])
            EX = None if ex_fly else [_buf(P, layers[l].out, dev, zero=False, dtype=sd, blocked=blk) for l in range(L)]
            cb = ChainBuilder(P, "POSENC", k8(self.E), site=("udf_tangent", _memo_token(self)))
            cb.posenc(x, net.multires, float(net.scale), tangent=d_g.contiguous())
            cb.init_store(R[0])
            if tn2:
                cb.absmax(amax[1:2])
            if bsc:
                cb.tile_scale(amax_out=TA)
            for l in range(L):
                pl = layers[l]
                nxt_skip = (l + 1) in self.skip
                if ex_fly:
                    cb.step("MULSP", pl.frag(_kind("fwd", "bwd")), k8(pl.inp), pl.out, X1=X[l + 1], C1=R[l + 1],
                            scale=self.inv_sqrt2 if nxt_skip else 1.0, xscale=self._xs(l),
                            pe_tail_col=self._skip_col(l + 1) if nxt_skip else -1, pe_tail_scale=self.inv_sqrt2,
                            pe_dst=R[l + 1] if nxt_skip else None)
                else:
                    cb.step("TANGENT", pl.frag(_kind("fwd", "bwd")), k8(pl.inp), pl.out, X1=X[l + 1], X2=DA[l], C1=R[l + 1],
                            C2=EX[l],
                            scale=self.inv_sqrt2 if nxt_skip else 1.0, xscale=self._xs(l),
                            pe_tail_col=self._skip_col(l + 1) if nxt_skip else -1, pe_tail_scale=self.inv_sqrt2,
                            pe_dst=R[l + 1] if nxt_skip else None)
            cb.launch()
        inv_scale = 1.0 / float(net.scale)
        if second and not head4_path:
            call("nudf_signed_colsum", ptr(sign), ptr(R[L]), R[L].shape[1], P, layers[L].inp, inv_scale, ptr(grads[L][0]))
        plL = layers[L]
        F = plL.out - 1
        ABAR = [None] * (L + 1)
        if head4_path:
            # the head's adjoint is [sign * d udf / scale | d feat]: d feat is used where it lies (tile load, GEMM operand)
            # and column 0 travels as a 4-wide operand -- no [P, 257] copy (52 us), no separate column-sum kernel (36 us)
            head4 = torch.empty(pad_rows(P), 4, device=dev)
            # (the second-order weight gradient's operand sign / scale comes out of the same launch)
            sg4 = torch.empty(pad_rows(P), 4, device=dev) if second else None
            if d_udf is not None:      # column 0 = sign * d udf / scale, the rest zero: one launch
                call("nudf_col0_seed4", ptr(sign), ptr(d_udf.reshape(-1).contiguous()), float(inv_scale), P, pad_rows(P),
                     ptr(head4), ptr(sg4))
            else:
                head4.zero_()
                if second:
                    call("nudf_col0_seed4", ptr(sign), None, float(inv_scale), P, pad_rows(P), ptr(sg4), None)
            r1, ldr1 = head4, 4
        else:
            ABAR[L] = _buf(P, plL.out, dev, zero=False)
            call("nudf_udf_head_bwd", ptr(sign), ptr(d_udf), ptr(d_feat), d_feat_ld, P, F, inv_scale, ptr(ABAR[L]),
                 ABAR[L].shape[1])
            r1, ldr1 = ABAR[L], ABAR[L].shape[1]
        for l in range(L):
            ABAR[l] = _buf(P, layers[l].out, dev, zero=False, dtype=X[L].dtype, blocked=_isblk(X[L]))
        # adjoint sweep: the tile starts as d feat (ABAR[L] columns 1..F); column 0 enters as a rank-1 term
        cb = ChainBuilder(P, "LOAD", k8(F), site=("udf_adjoint", _memo_token(self)))
        if d_feat is None:
            d_feat, d_feat_ld = torch.zeros(P, k8(F), device=dev), k8(F)
        cb.init_load(d_feat, d_feat_ld)
        if tn2:
            cb.absmax(amax[0:1])
        if bsc:
            cb.tile_scale(amax_in=TA)
        for l in range(L, 0, -1):
            pl = layers[l]
            sc = self.inv_sqrt2 if l in self.skip else 1.0
            x2 = (R[l] if ex_fly else EX[l - 1]) if second else None
            x3 = DA[l - 1] if ex_fly else None
            if l == L:
                cb.step("BWD", pl.frag(_kind("bwd_feat", "bwd")), k8(F), layers[l - 1].out, X1=X[l], X2=x2, X3=x3,
                        C1=ABAR[l - 1], r1_row=r1, ldr1=ldr1, r1_col=pl.W, scale=sc,
                        xscale=self._xs(l - 1))
            else:
                n_hid = layers[l - 1].out           # the skip layer's embedding columns carry no parameter gradient
                cb.step("BWD", pl.frag(_kind("bwd" if n_hid == pl.inp else "bwd_hid:%d" % n_hid, "bwd")), k8(pl.out),
                        n_hid, X1=X[l], X2=x2, X3=x3, C1=ABAR[l - 1], scale=sc, xscale=self._xs(l - 1))
        cb.launch()
        if grouped:
            # two grouped launches (adjoint pairs, then the second-order pairs accumulating into the same dW): each
            # is one resident wave of ~500 workgroups with 14-17 row chunks per tile instead of 128 -- 9x fewer
            # atomics per output element and 2 launch tails instead of 9 (1.67 -> 1.56 ms per step; ONE launch of all
            # 17 problems fills only 462 of the 512 slots and was slower).  The head layer enters as two problems:
            # rows 1.. of dW from d feat, row 0 from the 4-wide column-0 operand.
            dWL, dbL = grads[L]
            jobs = [(ABAR[l], layers[l].out, X[l], layers[l].in_pad, grads[l][0], grads[l][1]) for l in range(L)]
            if head4_path:
                assert d_feat.shape[1] == d_feat_ld and d_feat.is_contiguous()
                jobs.append((d_feat, F, X[L], plL.in_pad, dWL[1:], dbL[1:]))
                jobs.append((head4, 1, X[L], plL.in_pad, dWL[:1], dbL[:1]))
            else:
                jobs.append((ABAR[L], plL.out, X[L], plL.in_pad, dWL, dbL))
            # (f16x2: the A side -- ABAR, d feat, the head's column-0 operand -- carries the adjoint sweep's maximum, X is used as it is)
            gemm_tn_grouped(jobs, P, assign=assign, f16x2=tn2, amax_a=amax[0:1] if tn2 else None)
            if second:
                jobs = [(DA[l], layers[l].out, R[l], layers[l].in_pad, grads[l][0], None) for l in range(L)]
                if head4_path:
                    # d (row 0 of the head) through the d udf / dx path: sign^T R_L / scale
                    jobs.append((sg4, 1, R[L], plL.in_pad, dWL[:1], None))
                # (f16x2: the B side -- R -- carries the tangent sweep's maximum; DA and sign / scale are used as they are)
                gemm_tn_grouped(jobs, P, f16x2=tn2, amax_b=amax[1:2] if tn2 else None)
            return unpack_group(layers, grads, claim_grad_slot(self, layers))
        for l, pl in enumerate(layers):
            dW, db = grads[l]
            if second and l < L:
                gemm_tn(ABAR[l], pl.out, X[l], dW, pl.out, pl.in_pad, P, dbias=db, A2=DA[l], na2=pl.out, B2=R[l])
            else:
                gemm_tn(ABAR[l], pl.out, X[l], dW, pl.out, pl.in_pad, P, dbias=db)
        return unpack_group(layers, grads, claim_grad_slot(self, layers))

    def _forward_layers(self, x, need_grad_state, feat_ld=0, udf_only=False):
        """per-layer GEMM launches.  x [P,3] -> dict(udf [P], sign [P], feat [P, max(feat_ld, F)], state...).
        feat_ld > F additionally writes x into cols F..F+2 (the colour net's base-input layout)."""
        P = x.shape[0]
        dev = x.device
        L = self.L
        pack_group(self.layers)
        X = [_buf(P, pl.inp, dev) for pl in self.layers]
        self._embed(x, P, X)
        for l in range(L):
            pl = self.layers[l]
            sc = self.inv_sqrt2 if (l + 1) in self.skip else 1.0
            gemm_nn(X[l], pl.Wt, P, pl.out, pl.in_pad, "SOFTPLUS", C1=X[l + 1], bias=pl.bias, scale=sc)
        pl = self.layers[L]
        udf = torch.empty(P, device=dev)
        sign = torch.empty(P, device=dev) if need_grad_state else None
        F = pl.out - 1
        feat = None
        if not udf_only:
            ld = max(feat_ld, F)
            feat = (torch.zeros if ld > F else torch.empty)((P, ld), device=dev)
            if ld >= F + 3:
                call("nudf_copy_cols", ptr(x), 3, 1, ptr(feat) + 4 * F, ld, 3, P, 1.0)
        gemm_nn(X[L], pl.Wt, P, 1 if udf_only else pl.out, pl.in_pad, "UDFHEAD", C1=feat, C2=udf, ldc2=1, C3=sign,
                ldc3=1, bias=pl.bias, scale=1.0 / float(self.net.scale), iparam=self.head_type)
        return dict(udf=udf, sign=sign, feat=feat, X=X, P=P)

    def _xs(self, j):
        """softplus output of layer j is stored in X[j+1] divided by this factor (skip concat /sqrt(2))."""
        return math.sqrt(2.0) if (j + 1) in self.skip else 1.0

    def _gradient_layers(self, x, st):
        """reverse sweep for d udf / d x given forward state -> (g [P,3], DA list)."""
        P, L, dev = st["P"], self.L, x.device
        X = st["X"]
        DA = [_buf(P, self.layers[l].out, dev) for l in range(L)]
        plL = self.layers[L]
        call("nudf_udf_grad_seed", ptr(st["sign"]), ptr(plL.W), ptr(X[L]), X[L].shape[1], self._xs(L - 1), P,
             self.layers[L - 1].out, 1.0 / float(self.net.scale), ptr(DA[L - 1]), DA[L - 1].shape[1])
        Epad = pad32(self.E)
        demb_skip = None
        for l in range(L - 1, 0, -1):
            pl = self.layers[l]
            if l in self.skip:
                if demb_skip is not None:
                    raise NotImplementedError("more than one skip layer")
                demb_skip = torch.zeros(P, Epad, device=dev)
                gemm_nn(DA[l], pl.W, P, pl.inp, pl.out_pad, "SKIPSPLIT", C1=DA[l - 1], C2=demb_skip, X1=X[l],
                        iparam=self.layers[l - 1].out, scale=self.inv_sqrt2, xscale=self._xs(l - 1))
            else:
                gemm_nn(DA[l], pl.W, P, pl.inp, pl.out_pad, "MULSP", C1=DA[l - 1], X1=X[l], xscale=self._xs(l - 1))
        demb0 = torch.zeros(P, Epad, device=dev)
        pl0 = self.layers[0]
        gemm_nn(DA[0], pl0.W, P, pl0.inp, pl0.out_pad, "NONE", C1=demb0)
        g = torch.empty(P, 3, device=dev)
        call("nudf_posenc_vjp", ptr(x), 3, self.net.d_in, self.net.multires, float(self.net.scale), P,
             ptr(demb0), Epad, 1.0, ptr(demb_skip), Epad, 1.0, ptr(g))
        return g, DA

    def _backward_layers(self, x, st, DA, d_udf, d_feat, d_feat_ld, d_g):
        """-> list of parameter gradients in params() order.
        d_udf [P] / d_feat [P, F] (row stride d_feat_ld) / d_g [P,3]; any may be None."""
        P, L, dev = st["P"], self.L, x.device
        X, sign = st["X"], st["sign"]
        layers = self.layers
        grads = [pl.new_grad_buffers() for pl in layers]
        second = d_g is not None and DA is not None
        R = EX = None
        if second:
            # tangent sweep with direction d_g (forward-over-reverse; see DESIGN.md "second order")
            R = [_buf(P, pl.inp, dev) for pl in layers]
            EX = [_buf(P, layers[l].out, dev) for l in range(L)]
            self._embed(x, P, R, tangent=d_g.contiguous())
            for l in range(L):
                pl = layers[l]
                sc = self.inv_sqrt2 if (l + 1) in self.skip else 1.0
                gemm_nn(R[l], pl.Wt, P, pl.out, pl.in_pad, "TANGENT", C1=R[l + 1], C2=EX[l], X1=X[l + 1], X2=DA[l],
                        scale=sc, xscale=self._xs(l))
            # d W_L[0,:] += sum_p sign_p * R_L[p,:] / scale
            call("nudf_signed_colsum", ptr(sign), ptr(R[L]), R[L].shape[1], P, layers[L].inp,
                 1.0 / float(self.net.scale), ptr(grads[L][0]))
        # adjoint of the output layer
        plL = layers[L]
        ABAR = [None] * (L + 1)
        ABAR[L] = _buf(P, plL.out, dev)
        call("nudf_udf_head_bwd", ptr(sign), ptr(d_udf), ptr(d_feat), d_feat_ld, P, plL.out - 1,
             1.0 / float(self.net.scale), ptr(ABAR[L]), ABAR[L].shape[1])
        for l in range(L, 0, -1):
            pl = layers[l]
            ABAR[l - 1] = _buf(P, layers[l - 1].out, dev)
            sc = self.inv_sqrt2 if l in self.skip else 1.0
            gemm_nn(ABAR[l], pl.W, P, layers[l - 1].out, pl.out_pad, "BWD", C1=ABAR[l - 1], X1=X[l],
                    X2=EX[l - 1] if second else None, scale=sc, xscale=self._xs(l - 1))
        out = []
        for l, pl in enumerate(layers):
            dW, db = grads[l]
            if second and l < L:
                gemm_tn(ABAR[l], pl.out, X[l], dW, pl.out, pl.in_pad, P, dbias=db, A2=DA[l], na2=pl.out, B2=R[l])
            else:
                gemm_tn(ABAR[l], pl.out, X[l], dW, pl.out, pl.in_pad, P, dbias=db)
            out += pl.unpack_grads(dW, db)
        return out


# =========================================================================================
# ReLU chains (colour network, NeRF)
# =========================================================================================
def relu_chain_bwd(layers, inputs, outs, D_last, P, first_needs_input_grad, add_at=None):
    """Backward through `layers[i]: inputs[i] -> relu -> outs[i]` (the last layer's pre-activation
    gradient is D_last [P, out_pad]).  Returns (grads per layer, d_input of layer 0 or None).
    add_at = {layer_index: tensor} adds an extra adjoint to that layer's *output* before the mask."""
    n = len(layers)
    grads = alloc_grads(layers)
    D = D_last
    d_in0 = None
    tn_jobs = []
    for i in range(n - 1, -1, -1):
        pl = layers[i]
        tn_jobs.append((D, pl.out, inputs[i], pl.in_pad, grads[i][0], grads[i][1]))   # dW_i = D_i^T input_i, grouped below
        if i > 0:
            prev = layers[i - 1]
            Dn = _buf(P, prev.out, D.device)
            extra = add_at.get(i - 1) if add_at else None
            if extra is not None:
                gemm_nn(D, pl.W, P, prev.out, pl.out_pad, "ADDMASK", C1=Dn, X1=outs[i - 1], ldx1=outs[i - 1].shape[1],
                        X2=extra[0], ldx2=extra[1], x2_off=extra[2])
            else:
                gemm_nn(D, pl.W, P, prev.out, pl.out_pad, "MULMASK", C1=Dn, X1=outs[i - 1], ldx1=outs[i - 1].shape[1])
            D = Dn
        elif first_needs_input_grad:
            d_in0 = _buf(P, pl.inp, D.device, zero=False)    # pad columns are never read
            gemm_nn(D, pl.W, P, pl.inp, pl.out_pad, "NONE", C1=d_in0)
    gemm_tn_grouped(tn_jobs, P)
    return grads, d_in0


class ColorEngine:
    """ResidualRenderingNetwork: mode 'no_normal' (every shipped conf) and the reference's other branch (any other mode
    string, fields.py:456-461): the base input also carries the DETACHED unit normal and its negative."""

    def __init__(self, net):
        self.net = net
        n = net.num_layers - 1
        self.n = n
        F, H, dout = net.d_feature, net.d_hidden, net.d_out
        self.F, self.H, self.dout = F, H, dout
        self.npe = net.view_dim                        # width of PE(view_dirs)
        self.nrm = 0 if net.mode == "no_normal" else 6 # [normals 3 | -normals 3] behind the points
        # base input buffer [feat F | pts 3 (| n 3 | -n 3)]; reference order is [pts 3, (n 3, -n 3,) feat F]
        perm_b0 = [F + i for i in range(3 + self.nrm)] + list(range(F))
        # view input buffer [hidden H | PE(dir) | color_base dout]; reference order [PE(dir), color_base, hidden]
        perm_v0 = [H + i for i in range(self.npe)] + [H + self.npe + j for j in range(dout)] + list(range(H))
        self.base = [PackedLinear(getattr(net, f"lin_base{l}"), perm_b0 if l == 0 else None) for l in range(n)]
        self.view = [PackedLinear(getattr(net, f"lin{l}"), perm_v0 if l == 0 else None) for l in range(n)]

    def params(self):
        out = []
        for pl in self.view + self.base:               # registration order: lin*, then lin_base*
            out += pl.params()
        return out

    def invalidate(self):
        for pl in self.view + self.base:
            pl.invalidate()

    def mark_stale(self):
        for pl in self.view + self.base:
            pl._ver = None

    @property
    def cin_ld(self):
        return pad32(self.F + 3 + self.nrm)

    # -- dispatch: fused LDS-resident chains (default) or per-layer GEMM launches -------------------
    def _chain_ok(self):
        n = self.n
        ok = USE_CHAIN and n >= 2 and 2 * n <= CH_MAX_STEPS and self.H <= 256 and self.net.embedview_fn is not None
        ok = ok and k8(self.base[0].inp) <= 288 and k8(self.H + self.npe + self.dout) <= 288 and self.F % 4 == 0
        ok = ok and self.view[n - 1].out <= 32 and self.dout <= 32
        return ok

    def _kinds(self):
        kc = self.__dict__.setdefault("_kinds_cache", {})
        k = kc.get(_mode_key())
        if k is None:
            k = kc[_mode_key()] = tuple(tuple(dict.fromkeys(ks)) for ks in self._kinds_build())
        return k

    def _kinds_build(self):
        """fragment copies per layer, in the order base + view (the pack_group order)."""
        fw, bw = _kind("fwd", "fwd"), _kind("bwd", "bwd")
        kinds = [(fw, _kind("bwd_hid:%d" % self.F, "bwd"))]     # d CIN: only the feature columns carry a gradient
        kinds += [(fw, bw)] * (self.n - 1)
        kinds += [(fw, bw)] * self.n
        return kinds

    def forward(self, CIN, rays_d, S, P, keep_state=True, row_w=None):
        if self._chain_ok():
            return self._forward_chain(CIN, rays_d, S, P, keep_state, row_w)
        if row_w is not None:
            raise _lib.NudfError("the compositing sum inside the colour heads needs the fused chain launch")
        return self._forward_layers(CIN, rays_d, S, P, keep_state)

    def backward(self, st, color_base, color, d_cb, d_color, d_logits):
        if "chain" in st:
            return self._backward_chain(st, color_base, color, d_cb, d_color, d_logits)
        return self._backward_layers(st, color_base, color, d_cb, d_color, d_logits)

    def _forward_chain(self, CIN, rays_d, S, P, keep_state=True, row_w=None):
        """one launch: base branch (ReLU x4, sigmoid head) -> [hidden | PE(dir) | color_base] assembled in the LDS
        tile -> view branch (ReLU x4, sigmoid + logits head)   (fields.py:452-495).
        row_w [pad_rows(P)] (no-grad rendering, `keep_state` False): the compositing weight of every point -- the two
        sigmoid heads then return the per-32-point sums of weight x colour ([ceil(P / 32), 4] each) INSTEAD of the
        per-point colours, which never leave the chip (udf_renderer_blending.py:508-526 taken into the epilogue)."""
        dev, n = CIN.device, self.n
        fused = row_w is not None
        assert not (fused and keep_state)
        H, npe, dout = self.H, self.npe, self.dout
        pack_group(self.base + self.view, self._kinds())
        Pp = pad_rows(P)
        VIN = torch.empty(Pp, pad32(H + npe + dout), device=dev) if keep_state else None
        sd = _state_dtype()     # hidden activations: bf16 (4-point packed) in the 16-bit mode; CIN / VIN stay fp32
        HB = [CIN] + [_buf(P, H, dev, zero=False, dtype=sd) for _ in range(n - 1)] if keep_state else None
        HV = [VIN] + [_buf(P, H, dev, zero=False, dtype=sd) for _ in range(n - 1)] if keep_state else None
        cb = ChainBuilder(P, "LOAD", k8(self.base[0].inp), tile_rows=COLOR_TILE, site=("col_fwd", _memo_token(self)))
        cb.init_load(CIN, CIN.shape[1])
        cb.posenc(rays_d, self.net.multires_view, 1.0, x_div=S)
        for l in range(n - 1):
            pl = self.base[l]
            tap = (l == n - 2)      # hidden tap (post-ReLU) also feeds the view branch, then PE(dir) joins the tile
            cb.step("RELU", pl.frag(_kind("fwd", "fwd")), k8(pl.inp), pl.out, bias=pl.bias, C1=HB[l + 1] if keep_state else None,
                    C2=VIN if (tap and keep_state) else None, pe_tail_col=H if tap else -1, pe_tail_scale=1.0,
                    pe_dst=VIN if (tap and keep_state) else None)
        color_base = None if fused else torch.empty(Pp, dout, device=dev)
        sums_b = torch.empty(Pp // 32, 4, device=dev) if fused else None
        sums_c = torch.empty(Pp // 32, 4, device=dev) if fused else None
        pl = self.base[n - 1]
        cb.step("SIGMOIDN", pl.frag(_kind("fwd", "fwd")), k8(pl.inp), pl.out, bias=pl.bias, C1=color_base, C2=VIN if keep_state else None,
                c2_off=H + npe, iparam=dout, act_write=1, act_col0=H + npe, row_w=row_w, row_sums=sums_b)
        for l in range(n - 1):
            pl = self.view[l]
            cb.step("RELU", pl.frag(_kind("fwd", "fwd")), k8(pl.inp), pl.out, bias=pl.bias, C1=HV[l + 1] if keep_state else None)
        pl = self.view[n - 1]
        nb = pl.out - dout
        color = None if fused else torch.empty(Pp, dout, device=dev)
        logits = torch.empty(Pp, max(nb, 1), device=dev)
        cb.step("SIGMOIDN", pl.frag(_kind("fwd", "fwd")), k8(pl.inp), pl.out, bias=pl.bias, C1=color, C2=logits if nb > 0 else None,
                iparam=dout, act_write=0, row_w=row_w, row_sums=sums_c)
        cb.launch()
        if fused:
            return sums_b, sums_c, (logits[:P] if nb > 0 else None), None
        st = dict(HB=HB, HV=HV, P=P, chain=True) if keep_state else None
        return color_base[:P], color[:P], (logits[:P] if nb > 0 else None), st

    def _backward_chain(self, st, color_base, color, d_cb, d_color, d_logits):
        """view-branch and base-branch reverse sweeps as one launch each; all 10 weight gradients in one grouped GEMM."""
        P, n = st["P"], self.n
        HB, HV = st["HB"], st["HV"]
        dev = color.device
        H, npe, dout = self.H, self.npe, self.dout
        assign = tn_can_assign()
        grads = alloc_grads(self.view + self.base, zero=not assign)
        plv = self.view[n - 1]
        nb = plv.out - dout
        sd = HV[1].dtype        # adjoints of the hidden layers follow the saved activations (bf16 in the 16-bit mode)
        Dv = [_buf(P, H, dev, zero=False, dtype=sd) for _ in range(n - 1)] + [_buf(P, plv.out, dev, zero=False)]
        call("nudf_sigmoid_head_bwd", ptr(color), ptr(d_color), None, 0, dout, ptr(d_logits), max(nb, 1), nb, P,
             ptr(Dv[n - 1]), Dv[n - 1].shape[1])
        dVIN = _buf(P, self.view[0].inp, dev, zero=False)
        tn2 = TN_F16X2 and PRECISION == "bf16x3" and TN_SPLIT and not _is16(HV[1]) and COLOR_TILE in (0, 32, 64)
        amax = torch.zeros(1, device=dev) if tn2 else None      # max |Dv, Db| (NudfChain.absmax_out of both reverse sweeps)
        cb = ChainBuilder(P, "LOAD", k8(plv.out), tile_rows=COLOR_TILE, site=("col_bwd_view", _memo_token(self)))
        cb.init_load(Dv[n - 1], Dv[n - 1].shape[1])
        if tn2:
            cb.absmax(amax)
        bsc = _sweep_dtype("bwd") == "f16x2"          # per-tile scaling; d VIN of this sweep enters the next one as X2
        TV = torch.empty(pad_rows(P) // 32, device=dev) if bsc else None
        if bsc:
            cb.tile_scale(amax_out=TV)
        for i in range(n - 1, 0, -1):
            pl = self.view[i]
            cb.step("MULMASK", pl.frag(_kind("bwd", "bwd")), k8(pl.out), pl.inp, X1=HV[i], C1=Dv[i - 1])
        pl0 = self.view[0]
        cb.step("NONE", pl0.frag(_kind("bwd", "bwd")), k8(pl0.out), pl0.inp, C1=dVIN, act_write=0)
        cb.launch()
        # base head: d color_base = direct + through the view branch's input columns
        plb = self.base[n - 1]
        Db = [_buf(P, H, dev, zero=False, dtype=sd) for _ in range(n - 1)] + [_buf(P, plb.out, dev, zero=False)]
        call("nudf_sigmoid_head_bwd", ptr(color_base), ptr(d_cb), ptr(dVIN) + 4 * (H + npe), dVIN.shape[1],
             dout, None, 0, 0, P, ptr(Db[n - 1]), Db[n - 1].shape[1])
        dCIN = torch.empty(pad_rows(P), self.cin_ld, device=dev)
        cb = ChainBuilder(P, "LOAD", k8(plb.out), tile_rows=COLOR_TILE, site=("col_bwd_base", _memo_token(self)))
        cb.init_load(Db[n - 1], Db[n - 1].shape[1])
        if tn2:
            cb.absmax(amax)
        if bsc:
            cb.tile_scale(amax_in=TV)
        for i in range(n - 1, 0, -1):
            pl = self.base[i]
            if i == n - 1:   # the hidden tap's adjoint from the view branch joins before the ReLU mask
                cb.step("ADDMASK", pl.frag(_kind("bwd", "bwd")), k8(pl.out), pl.inp, X1=HB[i], X2=dVIN, C1=Db[i - 1])
            else:
                cb.step("MULMASK", pl.frag(_kind("bwd", "bwd")), k8(pl.out), pl.inp, X1=HB[i], C1=Db[i - 1])
        pb0 = self.base[0]
        cb.step("NONE", pb0.frag(_kind("bwd_hid:%d" % self.F, "bwd")), k8(pb0.out), self.F, C1=dCIN, act_write=0)
        cb.launch()
        jobs = []
        for i, pl in enumerate(self.view):
            jobs.append((Dv[i], pl.out, HV[i], pl.in_pad, grads[i][0], grads[i][1]))
        for i, pl in enumerate(self.base):
            jobs.append((Db[i], pl.out, HB[i], pl.in_pad, grads[n + i][0], grads[n + i][1]))
        gemm_tn_grouped(jobs, P, assign=assign, f16x2=tn2, amax_a=amax)
        return unpack_group(self.view + self.base, grads, claim_grad_slot(self, self.view + self.base)), dCIN[:P]

    def _forward_layers(self, CIN, rays_d, S, P, keep_state=True):
        """CIN [P, pad(F+3)] = [feature F | pts 3 | 0] (written by the UDF head + nudf_copy_cols)."""
        dev = CIN.device
        n = self.n
        pack_group(self.base + self.view)
        VIN = _zero_cols(torch.empty(P, pad32(self.H + self.npe + self.dout), device=dev), self.H + self.npe + self.dout)
        # PE(view_dirs) (fields.py:453-454); directions are per ray -> xdiv = S
        call("nudf_posenc", ptr(rays_d), 3, S, None, 3, self.net.multires_view, 1.0, P,
             ptr(VIN) + 4 * self.H, VIN.shape[1], 1.0, None, 0, 0.0)
        HB = [CIN]
        for l in range(n - 1):
            pl = self.base[l]
            h = _buf(P, pl.out, dev)
            if l == n - 2:   # hidden tap (post-ReLU) also feeds the view branch
                gemm_nn(HB[l], pl.Wt, P, pl.out, pl.in_pad, "RELU_DUAL", C1=h, C2=VIN, bias=pl.bias)
            else:
                gemm_nn(HB[l], pl.Wt, P, pl.out, pl.in_pad, "RELU", C1=h, bias=pl.bias)
            HB.append(h)
        color_base = torch.empty(P, self.dout, device=dev)
        pl = self.base[n - 1]
        gemm_nn(HB[n - 1], pl.Wt, P, pl.out, pl.in_pad, "SIGMOID", C1=color_base, C2=VIN, c2_off=self.H + self.npe,
                bias=pl.bias, iparam=self.dout)
        HV = [VIN]
        for l in range(n - 1):
            pl = self.view[l]
            h = _buf(P, pl.out, dev)
            gemm_nn(HV[l], pl.Wt, P, pl.out, pl.in_pad, "RELU", C1=h, bias=pl.bias)
            HV.append(h)
        pl = self.view[n - 1]
        color = torch.empty(P, self.dout, device=dev)
        nb = pl.out - self.dout
        logits = torch.empty(P, max(nb, 1), device=dev)
        gemm_nn(HV[n - 1], pl.Wt, P, pl.out, pl.in_pad, "SIGMOID", C1=color, C3=logits, bias=pl.bias,
                iparam=self.dout)
        st = dict(HB=HB, HV=HV, P=P) if keep_state else None
        return color_base, color, (logits if nb > 0 else None), st

    def _backward_layers(self, st, color_base, color, d_cb, d_color, d_logits):
        """-> (param grads in params() order, d_feat buffer [P, pad(F+3)] whose cols 0..F-1 are d feature)."""
        P, n = st["P"], self.n
        HB, HV = st["HB"], st["HV"]
        dev = color.device
        plv = self.view[n - 1]
        nb = plv.out - self.dout
        D = _buf(P, plv.out, dev, zero=False)           # nudf_sigmoid_head_bwd writes the pad columns
        call("nudf_sigmoid_head_bwd", ptr(color), ptr(d_color), None, 0, self.dout, ptr(d_logits), max(nb, 1), nb, P,
             ptr(D), D.shape[1])
        gv, dVIN = relu_chain_bwd(self.view, HV, HV[1:], D, P, True)
        # base head: d color_base = direct + through the view branch's input columns
        Db = _buf(P, self.dout, dev, zero=False)
        call("nudf_sigmoid_head_bwd", ptr(color_base), ptr(d_cb), ptr(dVIN) + 4 * (self.H + self.npe), dVIN.shape[1],
             self.dout, None, 0, 0, P, ptr(Db), Db.shape[1])
        gb, dCIN = relu_chain_bwd(self.base, HB, HB[1:], Db, P, True, add_at={n - 2: (dVIN, dVIN.shape[1], 0)})
        return unpack_group(self.view + self.base, gv + gb), dCIN[:P]


class PlainColorEngine:
    """RenderingNetwork (fields.py:325-397): one ReLU chain over an input assembled by the caller, sigmoid colour head
    (+ raw blending logits).  Not instantiated by any shipped conf (the runner uses the residual variant), so it runs on
    the per-layer GEMM launches: any input width / mode, no fused chain."""

    def __init__(self, net):
        self.net = net
        self.layers = [PackedLinear(getattr(net, f"lin{l}")) for l in range(net.num_layers - 1)]
        self.dout = net.d_out

    def params(self):
        out = []
        for pl in self.layers:
            out += pl.params()
        return out

    def forward(self, Xin, P, keep_state=True):
        """Xin [P, width] -> (out [P, last.out] with the first d_out columns through the sigmoid when squeeze_out)."""
        dev, n = Xin.device, len(self.layers)
        pack_group(self.layers)
        X = _buf(P, self.layers[0].inp, dev)
        X[:P, :Xin.shape[1]] = Xin
        H = [X]
        for l in range(n - 1):
            pl = self.layers[l]
            h = _buf(P, pl.out, dev)
            gemm_nn(H[l], pl.Wt, P, pl.out, pl.in_pad, "RELU", C1=h, bias=pl.bias)
            H.append(h)
        pl = self.layers[n - 1]
        nb = pl.out - self.dout
        color = torch.empty(P, self.dout, device=dev)
        logits = torch.empty(P, max(nb, 1), device=dev)
        if self.net.squeeze_out:
            gemm_nn(H[n - 1], pl.Wt, P, pl.out, pl.in_pad, "SIGMOID", C1=color, C3=logits, bias=pl.bias, iparam=self.dout)
        else:
            raw = _buf(P, pl.out, dev)
            gemm_nn(H[n - 1], pl.Wt, P, pl.out, pl.in_pad, "NONE", C1=raw, bias=pl.bias)
            color, logits = raw[:P, :self.dout].contiguous(), raw[:P, self.dout:pl.out].contiguous()
        st = dict(H=H, P=P) if keep_state else None
        return color, (logits if nb > 0 else None), st

    def backward(self, st, color, d_color, d_logits):
        """-> (param grads in params() order, d Xin [P, width])."""
        P, H = st["P"], st["H"]
        dev = color.device
        pl = self.layers[-1]
        nb = pl.out - self.dout
        D = _buf(P, pl.out, dev, zero=True)
        if self.net.squeeze_out:
            if d_color is None:
                d_color = torch.zeros_like(color)
            if nb > 0 and d_logits is None:
                d_logits = torch.zeros(P, nb, device=dev)
            call("nudf_sigmoid_head_bwd", ptr(color), ptr(d_color.contiguous()), None, 0, self.dout,
                 ptr(d_logits.contiguous()) if nb > 0 else None, max(nb, 1), nb, P, ptr(D), D.shape[1])
        else:
            if d_color is not None:
                D[:P, :self.dout] = d_color
            if nb > 0 and d_logits is not None:
                D[:P, self.dout:pl.out] = d_logits
        grads, dX = relu_chain_bwd(self.layers, H, H[1:], D, P, True)
        return unpack_group(self.layers, grads), dX[:P, :self.layers[0].inp]


class NerfEngine:
    """Background NeRF with view directions (fields.py:541-628)."""

    def __init__(self, net):
        self.net = net
        # (use_viewdirs=False never gets here: the reference's forward asserts False for it, and so does models.fields.NeRF)
        assert net.use_viewdirs, "NeRF(use_viewdirs=False) has no forward in the reference (fields.py:629-630)"
        self.D, self.W = net.D, net.W
        self.e = net.input_ch
        self.ev = net.input_ch_view
        self.skips = set(net.skips)
        self.pts = []
        for i in range(self.D):
            lin = net.pts_linears[i]
            perm = None
            if (i - 1) in self.skips:   # input was cat([input_pts e, h W]); buffer is [h W | e]
                perm = [self.W + j for j in range(self.e)] + list(range(self.W))
            self.pts.append(PackedLinear(lin, perm))
        self.alpha = PackedLinear(net.alpha_linear)
        self.feature = PackedLinear(net.feature_linear)
        self.views = PackedLinear(net.views_linears[0])
        self.rgb = PackedLinear(net.rgb_linear)
        if len(net.views_linears) != 1:
            raise NotImplementedError("NeRF with more than one views layer")

    def _all(self):
        # registration order of nn.Module: pts_linears, views_linears, feature_linear, alpha_linear, rgb_linear
        return self.pts + [self.views, self.feature, self.alpha, self.rgb]

    def params(self):
        out = []
        for pl in self._all():
            out += pl.params()
        return out

    def invalidate(self):
        for pl in self._all():
            pl.invalidate()

    def mark_stale(self):
        for pl in self._all():
            pl._ver = None

    # -- dispatch: fused LDS-resident chains (default) or per-layer GEMM launches -------------------
    def _skip_layer(self):
        """index of the pts layer whose input is cat([input_pts, h]) (fields.py:607-609), or -1."""
        sk = [i + 1 for i in sorted(self.skips) if i + 1 < self.D]
        return sk[0] if sk else -1

    def _chain_ok(self):
        net = self.net
        sk = [i for i in self.skips if 0 <= i < self.D]
        ok = USE_CHAIN and os.environ.get("NUDF_NERF_CHAIN", "1") != "0"      # A/B switch for profiling
        ok = ok and self.W <= 256 and self.W % 32 == 0 and self.D + 5 <= CH_MAX_STEPS and len(sk) <= 1
        ok = ok and (not sk or sk[0] < self.D - 1) and k8(self.e) <= 288 and k8(self.W + self.ev) <= 288
        ok = ok and net.d_in_view == 3 and self.ev == 3 * (2 * net.multires_view + 1) and self.views.out <= 256
        return ok

    def _kinds(self):
        kc = self.__dict__.setdefault("_kinds_cache", {})
        k = kc.get(_mode_key())
        if k is None:
            k = kc[_mode_key()] = tuple(tuple(dict.fromkeys(ks)) for ks in self._kinds_build())
        return k

    def _kinds_build(self):
        """fragment copies per layer in _all() order (pts, views, feature, alpha, rgb)."""
        fw, bw = _kind("fwd", "fwd"), _kind("bwd", "bwd")
        W, e, j = self.W, self.e, self._skip_layer()
        kinds = []
        for i in range(self.D):
            if i == j:
                kinds.append((_kind("fwd_in:0:%d" % W, "fwd"), _kind("fwd_in:%d:%d" % (W, e), "fwd"),
                              _kind("bwd_hid:%d" % W, "bwd")))
            else:
                kinds.append((fw,) if i == 0 else (fw, bw))
        kinds += [(fw, _kind("bwd_hid:%d" % W, "bwd")), (fw, bw), (fw,), (fw, bw)]
        return kinds

    def forward(self, pts4, rays_d, S, P, keep_state=True):
        if self._chain_ok():
            return self._forward_chain(pts4, rays_d, S, P, keep_state)
        return self._forward_layers(pts4, rays_d, S, P, keep_state)

    def backward(self, st, d_sigma, d_rgb):
        if "chain" in st:
            return self._backward_chain(st, d_sigma, d_rgb)
        return self._backward_layers(st, d_sigma, d_rgb)

    def _forward_chain(self, pts4, rays_d, S, P, keep_state=True):
        """the whole background network (fields.py:599-628) as one launch: PE(pts) tile -> 8 ReLU layers (the skip
        layer's cat([input_pts, h]) is 340 wide, more than the 288-column LDS tile, so its PE part is multiplied by
        an extra step while the tile still holds PE(pts) and joins through the RELUADD epilogue) -> density head,
        feature layer + PE(dir) assembled in the tile -> view layer -> colour head."""
        dev, D, W, e, ev = pts4.device, self.D, self.W, self.e, self.ev
        net = self.net
        fw = _kind("fwd", "fwd")
        j = self._skip_layer()
        pack_group(self._all(), self._kinds())
        Hin = [_buf(P, self.pts[0].inp, dev)]
        if keep_state:
            Hin += [_buf(P, pl.inp, dev) for pl in self.pts[1:]]
        d2 = Hin[j] if (j >= 0 and keep_state) else None
        call("nudf_posenc", ptr(pts4), net.d_in, 1, None, net.d_in, net.multires, 1.0, P,
             ptr(Hin[0]), Hin[0].shape[1], 1.0,
             (ptr(d2) + 4 * W) if d2 is not None else None, d2.shape[1] if d2 is not None else 0, 1.0)
        Pp = pad_rows(P)
        h_last = _buf(P, W, dev, zero=False) if keep_state else None
        VIN = _zero_cols(torch.empty(Pp, pad32(W + ev), device=dev), W + ev) if keep_state else None
        hv = _buf(P, self.views.out, dev, zero=False) if keep_state else None
        sigma = torch.empty(Pp, 1, device=dev)
        rgb = torch.empty(Pp, 3, device=dev)
        cb = ChainBuilder(P, "LOAD", k8(e), site=("nerf_fwd", _memo_token(self)))
        cb.init_load(Hin[0], Hin[0].shape[1])
        cb.posenc(rays_d, net.multires_view, 1.0, x_div=S)
        SK = None
        if j >= 0:
            SK = _buf(P, W, dev, zero=False)
            cb.step("NONE", self.pts[j].frag(_kind("fwd_in:%d:%d" % (W, e), "fwd")), k8(e), W, C1=SK, act_write=0)
        for i in range(D):
            pl = self.pts[i]
            dst = None
            if keep_state:
                dst = Hin[i + 1] if i + 1 < D else h_last
            if i == j:
                cb.step("RELUADD", pl.frag(_kind("fwd_in:0:%d" % W, "fwd")), k8(W), W, bias=pl.bias, X2=SK, C1=dst)
            else:
                cb.step("RELU", pl.frag(fw), k8(pl.inp), pl.out, bias=pl.bias, C1=dst)
        cb.step("NONE", self.alpha.frag(fw), k8(W), 1, bias=self.alpha.bias, C1=sigma, act_write=0)
        cb.step("NONE", self.feature.frag(fw), k8(W), W, bias=self.feature.bias, C1=VIN, pe_tail_col=W, pe_tail_scale=1.0,
                pe_dst=VIN)
        cb.step("RELU", self.views.frag(fw), k8(W + ev), self.views.out, bias=self.views.bias, C1=hv)
        cb.step("NONE", self.rgb.frag(fw), k8(self.views.out), 3, bias=self.rgb.bias, C1=rgb, act_write=0)
        cb.launch()
        st = dict(Hin=Hin, h_last=h_last, VIN=VIN, hv=hv, P=P, chain=True) if keep_state else None
        return sigma[:P], rgb[:P], st

    def _backward_chain(self, st, d_sigma, d_rgb):
        """one reverse sweep (colour head -> view layer -> feature layer joined with the density head's rank-1 adjoint
        -> pts layers 7..1) and one grouped GEMM for all 12 weight gradients."""
        P, D, W = st["P"], self.D, self.W
        Hin, h_last, VIN, hv = st["Hin"], st["h_last"], st["VIN"], st["hv"]
        dev = VIN.device
        bw = _kind("bwd", "bwd")
        j = self._skip_layer()
        layers = self._all()
        assign = tn_can_assign()
        grads = alloc_grads(layers, zero=not assign)
        Pp = pad_rows(P)
        Drgb = torch.zeros(Pp, 32, device=dev)
        call("nudf_copy_cols", ptr(d_rgb), 3, 1, ptr(Drgb), 32, 3, P, 1.0)
        Dsig = torch.zeros(Pp, 32, device=dev)
        call("nudf_copy_cols", ptr(d_sigma), 1, 1, ptr(Dsig), 32, 1, P, 1.0)
        Dv = _buf(P, self.views.out, dev, zero=False)
        dF = _buf(P, W, dev, zero=False)
        Dp = [_buf(P, W, dev, zero=False) for _ in range(D)]
        cb = ChainBuilder(P, "LOAD", k8(3), site=("nerf_bwd", _memo_token(self)))
        cb.init_load(Drgb, Drgb.shape[1])
        if _sweep_dtype("bwd") == "f16x2":
            cb.tile_scale()
        cb.step("MULMASK", self.rgb.frag(bw), k8(3), self.views.out, X1=hv, C1=Dv)
        cb.step("NONE", self.views.frag(_kind("bwd_hid:%d" % W, "bwd")), k8(self.views.out), W, C1=dF)
        cb.step("MULMASK", self.feature.frag(bw), k8(W), W, X1=h_last, r1_row=Dsig, ldr1=Dsig.shape[1],
                r1_col=self.alpha.W, C1=Dp[D - 1])
        for i in range(D - 1, 0, -1):
            pl = self.pts[i]
            kind = _kind("bwd_hid:%d" % W, "bwd") if i == j else bw
            cb.step("MULMASK", pl.frag(kind), k8(pl.out), W, X1=Hin[i], C1=Dp[i - 1], act_write=1 if i > 1 else 0)
        cb.launch()
        jobs = [(Dp[i], pl.out, Hin[i], pl.in_pad, grads[i][0], grads[i][1]) for i, pl in enumerate(self.pts)]
        jobs.append((Dv, self.views.out, VIN, self.views.in_pad, grads[D][0], grads[D][1]))
        jobs.append((dF, self.feature.out, h_last, self.feature.in_pad, grads[D + 1][0], grads[D + 1][1]))
        jobs.append((Dsig, 1, h_last, self.alpha.in_pad, grads[D + 2][0], grads[D + 2][1]))
        jobs.append((Drgb, 3, hv, self.rgb.in_pad, grads[D + 3][0], grads[D + 3][1]))
        gemm_tn_grouped(jobs, P, assign=assign)
        return unpack_group(layers, grads, claim_grad_slot(self, layers))

    def _forward_layers(self, pts4, rays_d, S, P, keep_state=True):
        dev = pts4.device
        pack_group(self._all())
        Hin = [_buf(P, pl.inp, dev) for pl in self.pts]
        skip_layers = [i + 1 for i in sorted(self.skips) if i + 1 < self.D]
        d2 = Hin[skip_layers[0]] if skip_layers else None
        call("nudf_posenc", ptr(pts4), self.net.d_in, 1, None, self.net.d_in, self.net.multires, 1.0, P,
             ptr(Hin[0]), Hin[0].shape[1], 1.0,
             (ptr(d2) + 4 * self.W) if d2 is not None else None, d2.shape[1] if d2 is not None else 0, 1.0)
        if len(skip_layers) > 1:
            raise NotImplementedError("more than one NeRF skip")
        outs = []
        for i in range(self.D):
            pl = self.pts[i]
            dst = Hin[i + 1] if i + 1 < self.D else _buf(P, pl.out, dev)
            gemm_nn(Hin[i], pl.Wt, P, pl.out, pl.in_pad, "RELU", C1=dst, bias=pl.bias)
            outs.append(dst)
        h = outs[-1]
        if (self.D - 1) in self.skips:
            raise NotImplementedError("skip after the last NeRF layer")
        sigma = torch.empty(P, 1, device=dev)
        gemm_nn(h, self.alpha.Wt, P, 1, self.alpha.in_pad, "NONE", C1=sigma, bias=self.alpha.bias)
        VIN = torch.zeros(P, pad32(self.W + self.ev), device=dev)
        gemm_nn(h, self.feature.Wt, P, self.feature.out, self.feature.in_pad, "NONE", C1=VIN, bias=self.feature.bias)
        call("nudf_posenc", ptr(rays_d), 3, S, None, self.net.d_in_view, self.net.multires_view, 1.0, P,
             ptr(VIN) + 4 * self.W, VIN.shape[1], 1.0, None, 0, 0.0)
        hv = _buf(P, self.views.out, dev)
        gemm_nn(VIN, self.views.Wt, P, self.views.out, self.views.in_pad, "RELU", C1=hv, bias=self.views.bias)
        rgb = torch.empty(P, 3, device=dev)
        gemm_nn(hv, self.rgb.Wt, P, 3, self.rgb.in_pad, "NONE", C1=rgb, bias=self.rgb.bias)
        st = dict(Hin=Hin, outs=outs, VIN=VIN, hv=hv, P=P) if keep_state else None
        return sigma, rgb, st

    def _backward_layers(self, st, d_sigma, d_rgb):
        P = st["P"]
        Hin, outs, VIN, hv = st["Hin"], st["outs"], st["VIN"], st["hv"]
        dev = VIN.device
        # rgb head
        Drgb = torch.zeros(P, 32, device=dev)
        call("nudf_copy_cols", ptr(d_rgb), 3, 1, ptr(Drgb), 32, 3, P, 1.0)
        g_rgb = self.rgb.new_grad_buffers()
        gemm_tn(Drgb, 3, hv, g_rgb[0], 3, self.rgb.in_pad, P, dbias=g_rgb[1])
        Dv = _buf(P, self.views.out, dev)
        gemm_nn(Drgb, self.rgb.W, P, self.views.out, self.rgb.out_pad, "MULMASK", C1=Dv, X1=hv)
        g_views = self.views.new_grad_buffers()
        gemm_tn(Dv, self.views.out, VIN, g_views[0], self.views.out, self.views.in_pad, P, dbias=g_views[1])
        dVIN = _buf(P, self.views.inp, dev, zero=False)  # only its first `feature.out` columns are read
        gemm_nn(Dv, self.views.W, P, self.views.inp, self.views.out_pad, "NONE", C1=dVIN)
        # feature / alpha heads share h = outs[-1]
        g_feat = self.feature.new_grad_buffers()
        gemm_tn(dVIN, self.feature.out, outs[-1], g_feat[0], self.feature.out, self.feature.in_pad, P, dbias=g_feat[1])
        Dsig = torch.zeros(P, 32, device=dev)
        call("nudf_copy_cols", ptr(d_sigma), 1, 1, ptr(Dsig), 32, 1, P, 1.0)
        g_alpha = self.alpha.new_grad_buffers()
        gemm_tn(Dsig, 1, outs[-1], g_alpha[0], 1, self.alpha.in_pad, P, dbias=g_alpha[1])
        T1 = _buf(P, self.W, dev)
        gemm_nn(Dsig, self.alpha.W, P, self.W, self.alpha.out_pad, "NONE", C1=T1)
        Dh = _buf(P, self.W, dev)
        gemm_nn(dVIN, self.feature.W, P, self.W, self.feature.out_pad, "ADDMASK", C1=Dh, X1=outs[-1],
                ldx1=outs[-1].shape[1], X2=T1)
        gp, _ = relu_chain_bwd(self.pts, Hin, outs, Dh, P, False)
        return unpack_group(self._all(), gp + [g_views, g_feat, g_alpha, g_rgb])
