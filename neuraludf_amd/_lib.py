"""ctypes binding of libnudf.so (declared in include/nudf.h).

The product path has NO fallback: if the library is missing or a call fails this raises.
torch is imported first so that the library's HIP runtime dependency (libamdhip64.so.7)
resolves to the copy already loaded by PyTorch-ROCm (one HIP runtime per process)."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before libnudf, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NUDF_LIB") or os.path.join(_HERE, "libnudf.so")      # NUDF_LIB: A/B builds of the library
ABI_VERSION = 105         # nudf_version() of the include/nudf.h these ctypes structures mirror

c_fp = C.c_void_p
i32 = C.c_int32
f32 = C.c_float


class NudfError(RuntimeError):
    pass


class GemmNN(C.Structure):
    _fields_ = [("A", c_fp), ("lda", i32), ("B", c_fp), ("ldb", i32), ("bias", c_fp),
                ("C1", c_fp), ("ldc1", i32), ("C2", c_fp), ("ldc2", i32), ("C3", c_fp), ("ldc3", i32),
                ("X1", c_fp), ("ldx1", i32), ("X2", c_fp), ("ldx2", i32),
                ("M", i32), ("N", i32), ("K", i32), ("epi", i32), ("iparam", i32), ("scale", f32), ("xscale", f32)]


class GemmTN(C.Structure):
    _fields_ = [("A1", c_fp), ("lda1", i32), ("na1", i32), ("B1", c_fp), ("ldb1", i32),
                ("A2", c_fp), ("lda2", i32), ("na2", i32), ("B2", c_fp), ("ldb2", i32),
                ("C", c_fp), ("ldc", i32), ("dbias", c_fp), ("M", i32), ("NA", i32), ("NB", i32),
                ("rows_per_block", i32), ("prec", i32)]


TN_MAX_PROBLEMS = 12


class GemmTNProblem(C.Structure):
    _fields_ = [("A1", c_fp), ("B1", c_fp), ("C", c_fp), ("dbias", c_fp), ("lda1", i32), ("ldb1", i32), ("ldc", i32),
                ("NA", i32), ("NB", i32), ("tile_start", i32), ("flags", i32)]


class GemmTNGroup(C.Structure):
    _fields_ = [("n_problems", i32), ("M", i32), ("rows_per_block", i32), ("total_tiles", i32), ("prec", i32),
                ("prob", GemmTNProblem * TN_MAX_PROBLEMS), ("workspace", c_fp), ("workspace_floats", C.c_int64),
                ("assign", i32), ("amax_a", c_fp), ("amax_b", c_fp)]


class BlendLoss(C.Structure):
    _fields_ = [("cb", c_fp), ("c", c_fp), ("pix", c_fp), ("gt", c_fp), ("err", c_fp), ("patch_mask", c_fp),
                ("weight_sum", c_fp), ("err_sorted", c_fp), ("order", c_fp), ("m", c_fp), ("err_masked", c_fp), ("sums", c_fp),
                ("sums_ws", c_fp), ("w_dev", c_fp), ("out", c_fp), ("d_total", c_fp), ("d_cb", c_fp), ("d_c", c_fp),
                ("d_pix", c_fp), ("d_err", c_fp), ("d_sums", c_fp), ("N", i32), ("sums_nblk", i32), ("n_rays", f32),
                ("trim_ratio", f32)]


class RayBatch(C.Structure):
    _fields_ = [("image", c_fp), ("mask", c_fp), ("intrinsics_inv", c_fp), ("pose", c_fp), ("pixels_x", c_fp),
                ("pixels_y", c_fp), ("N", i32), ("H", i32), ("W", i32), ("h_patch_size", i32), ("rays", c_fp),
                ("ndc_uv", c_fp), ("xyz_cam", c_fp), ("near", c_fp), ("far", c_fp), ("patch_color", c_fp),
                ("patch_mask", c_fp)]


class Composite(C.Structure):
    _fields_ = [("rays_o", c_fp), ("rays_d", c_fp), ("z", c_fp), ("udf", c_fp), ("grad", c_fp),
                ("color", c_fp), ("color_base", c_fp), ("bg_z", c_fp), ("bg_sigma", c_fp), ("bg_color", c_fp),
                ("scal", c_fp), ("sample_dist", c_fp), ("background_rgb", c_fp),
                ("N", i32), ("S", i32), ("n_out", i32), ("s_nominal", i32),
                ("has_anneal", i32), ("cos_anneal", f32), ("flip_saturation", f32),
                ("use_norm_grad", i32), ("sparse_scale", f32), ("alpha_type", i32),
                ("weights", c_fp), ("out_color", c_fp), ("out_color_base", c_fp), ("out_depth", c_fp),
                ("out_normals", c_fp), ("out_wsum", c_fp), ("out_wsum_all", c_fp), ("sums", c_fp), ("ws", c_fp),
                ("o_alpha", c_fp), ("o_alpha_plus", c_fp), ("o_alpha_minus", c_fp), ("o_vis_prob", c_fp),
                ("o_alpha_occ", c_fp), ("o_raw_occ", c_fp), ("o_true_cos", c_fp), ("o_grad_mag", c_fp),
                ("o_mid_z", c_fp), ("o_dists", c_fp), ("o_inside", c_fp), ("o_flip", c_fp), ("sched", c_fp),
                ("p_variance", c_fp), ("p_beta", c_fp), ("p_gamma", c_fp), ("beta_hi", f32), ("defer_sums", i32),
                ("scal_out", c_fp), ("recip_out", c_fp)]


class CompositeGrad(C.Structure):
    _fields_ = [("d_color", c_fp), ("d_color_base", c_fp), ("d_weights", c_fp), ("d_depth", c_fp),
                ("d_normals", c_fp), ("d_wsum", c_fp), ("d_wsum_all", c_fp), ("d_sums", c_fp),
                ("o_d_udf", c_fp), ("o_d_grad", c_fp), ("o_d_color", c_fp), ("o_d_color_base", c_fp),
                ("o_d_bg_sigma", c_fp), ("o_d_bg_color", c_fp), ("o_d_scal", c_fp), ("ws", c_fp), ("o_d_param", c_fp)]


class Upsample(C.Structure):
    _fields_ = [("rays_o", c_fp), ("rays_d", c_fp), ("z", c_fp), ("udf", c_fp), ("u", c_fp),
                ("sample_dist", c_fp), ("gamma_dev", c_fp),
                ("N", i32), ("M", i32), ("K", i32), ("mode", i32),
                ("inv_s", f32), ("beta", f32), ("gamma", f32),
                ("z_new", c_fp), ("pts_new", c_fp), ("dbg", c_fp),
                ("prev_z", c_fp), ("prev_udf", c_fp), ("add_z", c_fp), ("add_udf", c_fp),
                ("z_merged", c_fp), ("udf_merged", c_fp), ("merge_K", i32)]


class PixelBlend(C.Structure):
    _fields_ = [("pts", c_fp), ("logits", c_fp), ("nl", i32), ("proj", c_fp), ("imgs", c_fp),
                ("P", i32), ("V", i32), ("H", i32), ("W", i32), ("pix", c_fp), ("img_layout", i32)]


class PixelComposite(C.Structure):
    _fields_ = [("w", c_fp), ("pix", c_fp), ("pts", c_fp), ("bg_in", c_fp), ("bg_tail", c_fp),
                ("N", i32), ("S", i32), ("n_out", i32), ("out", c_fp)]


class PatchBlend(C.Structure):
    _fields_ = [("pts", c_fp), ("grad", c_fp), ("rays_d", c_fp), ("uv", c_fp), ("logits", c_fp), ("nl", i32),
                ("w", c_fp), ("ldw", i32), ("ref_cam", c_fp), ("src_cam", c_fp), ("imgs", c_fp),
                ("N", i32), ("S", i32), ("V", i32), ("H", i32), ("W", i32), ("hps", i32),
                ("patch_colors", c_fp), ("patch_mask", c_fp), ("img_layout", i32)]


class PatchWarp(C.Structure):
    _fields_ = [("pts", c_fp), ("normals", c_fp), ("uv", c_fp), ("ref_cam", c_fp), ("src_cam", c_fp), ("imgs", c_fp),
                ("N", i32), ("S", i32), ("V", i32), ("H", i32), ("W", i32), ("hps", i32), ("img_layout", i32),
                ("colors", c_fp), ("mask", c_fp)]


ADAM_MAX_TENSORS = 64
ADAM_MAX_GROUPS = 4


class AdamTensor(C.Structure):
    _fields_ = [("p", c_fp), ("g", c_fp), ("m", c_fp), ("v", c_fp), ("n", i32), ("group", i32), ("neg_step_size", f32), ("bc2_sqrt", f32)]


class AdamGroup(C.Structure):
    _fields_ = [("one_minus_beta1", f32), ("beta2", f32), ("one_minus_beta2", f32), ("eps", f32)]


class Adam(C.Structure):
    _fields_ = [("n_tensors", i32), ("pad_", i32), ("t", AdamTensor * ADAM_MAX_TENSORS),
                ("block_start", i32 * (ADAM_MAX_TENSORS + 1)), ("pad2_", i32), ("group", AdamGroup * ADAM_MAX_GROUPS),
                ("dyn", c_fp)]


CH_MAX_STEPS = 14
CH_STATE16 = 32     # NudfChainStep.layout: the step's stored-state arrays hold bf16
CH_P4_X1, CH_P4_C1 = 64, 128   # ReLU-family steps of the 16-bit mode: that array is bf16, 4-point packed
CH = dict(NONE=0, SOFTPLUS=1, MULSP=2, TANGENT=3, BWD=4, UDFHEAD=5, RELU=6, SIGMOIDN=7, MULMASK=8, ADDMASK=9, RELUADD=10)
CH_INIT = dict(LOAD=0, POSENC=1, SEED=2)


class ChainStep(C.Structure):
    _fields_ = [("Bp", c_fp), ("bias", c_fp), ("X1", c_fp), ("X2", c_fp), ("C1", c_fp), ("C2", c_fp),
                ("r1_row", c_fp), ("r1_col", c_fp), ("K", i32), ("N", i32), ("epi", i32), ("iparam", i32),
                ("ldx1", i32), ("ldx2", i32), ("ldc1", i32), ("ldc2", i32), ("ldr1", i32), ("prec", i32), ("act_write", i32),
                ("act_col0", i32), ("pe_tail_col", i32), ("ld_pe", i32), ("pe_dst", c_fp), ("pe_tail_scale", f32),
                ("scale", f32), ("xscale", f32), ("layout", i32), ("row_w", c_fp), ("row_sums", c_fp), ("X3", c_fp),
                ("ldx3", i32)]


class Chain(C.Structure):
    _fields_ = [("P", i32), ("n_steps", i32), ("init", i32), ("k0", i32), ("x_div", i32), ("tile_rows", i32), ("lda0", i32),
                ("ldg0", i32), ("pe_L", i32), ("pe_jvp", i32), ("init_state16", i32), ("pe_in_scale", f32), ("seed_scale", f32),
                ("seed_xscale", f32), ("A0", c_fp), ("G0", c_fp), ("x", c_fp), ("v", c_fp), ("seed_sign", c_fp),
                ("seed_wrow", c_fp), ("dbg", c_fp), ("absmax_out", c_fp), ("tile_amax_in", c_fp),
                ("tile_amax_out", c_fp), ("tile_scale", C.c_int32), ("reserved0", C.c_int32), ("step", ChainStep * CH_MAX_STEPS)]


PACK_MAX_LAYERS = 16
PACK_MAX_FRAGS = 4


class PackFrag(C.Structure):
    _fields_ = [("dst", c_fp), ("transpose", i32), ("o0", i32), ("i0", i32), ("K", i32), ("N", i32), ("dtype", i32)]


class PackLayer(C.Structure):
    _fields_ = [("v", c_fp), ("g", c_fp), ("perm", c_fp), ("W", c_fp), ("Wt", c_fp), ("inv_norm", c_fp),
                ("out", i32), ("in_", i32), ("ldw", i32), ("ldwt", i32), ("nfrag", i32), ("row_start", i32),
                ("frag", PackFrag * PACK_MAX_FRAGS)]


class PackMulti(C.Structure):
    _fields_ = [("n_layers", i32), ("total_rows", i32), ("layer", PackLayer * PACK_MAX_LAYERS)]


class UnpackLayer(C.Structure):
    _fields_ = [("dW", c_fp), ("v", c_fp), ("g", c_fp), ("inv_norm", c_fp), ("perm", c_fp), ("dv", c_fp),
                ("dg", c_fp), ("db_in", c_fp), ("db_out", c_fp), ("out", i32), ("in_", i32), ("ldw", i32), ("row_start", i32)]


class UnpackMulti(C.Structure):
    _fields_ = [("n_layers", i32), ("total_rows", i32), ("layer", UnpackLayer * PACK_MAX_LAYERS)]


# float offsets of the device loss-weight vector (include/nudf.h NUDF_LW_*)
LW = dict(color_base=0, color=1, color_pixel=2, color_patch=3, igr=4, igr_ns=5, sparse=6, mask=7, color_sum=8)
LW_COUNT = 16

EPI = dict(NONE=0, SOFTPLUS=1, RELU=2, MUL=3, MULMASK=4, TANGENT=5, BWD=6, SIGMOID=7, UDFHEAD=8,
           SKIPSPLIT=9, RELU_DUAL=10, ADDMASK=11, MULSP=12)

# every symbol include/nudf.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "nudf_version", "nudf_last_error", "nudf_set_status_flag", "nudf_status_flag", "nudf_gemm_nn", "nudf_set_gemm_variant", "nudf_gemm_tn", "nudf_gemm_tn_grouped", "nudf_gemm_tn_grouped_workspace", "nudf_gemm_tn_grouped_plan", "nudf_set_tn_flags", "nudf_patch_metric", "nudf_set_tn_debug", "nudf_composite_fwd",
    "nudf_composite_bwd", "nudf_partial_sums", "nudf_composite_colour_finish", "nudf_set_composite_blocked", "nudf_upsample", "nudf_merge", "nudf_merge_points", "nudf_coarse_z", "nudf_coarse_start", "nudf_outside_z",
    "nudf_ray_points", "nudf_posenc", "nudf_posenc_vjp", "nudf_copy_cols", "nudf_add_cols",
    "nudf_udf_grad_seed", "nudf_udf_head_bwd", "nudf_signed_colsum", "nudf_sigmoid_head_bwd",
    "nudf_weightnorm_pack", "nudf_weightnorm_unpack_grad",
    "nudf_pixel_blend_fwd", "nudf_pixel_blend_bwd", "nudf_pixel_composite_fwd", "nudf_pixel_composite_bwd",
    "nudf_patch_blend_fwd", "nudf_patch_blend_bwd", "nudf_ssim_patch", "nudf_pixel_warp", "nudf_patch_warp",
    "nudf_adam_step", "nudf_adam_chunk", "nudf_mlp_chain", "nudf_set_chain_t16", "nudf_pack_frag",
    "nudf_weightnorm_pack_multi", "nudf_weightnorm_unpack_grad_multi",
    "nudf_scalars_fwd", "nudf_scalars_bwd", "nudf_l1_sum_fwd", "nudf_l1_sum_bwd",
    "nudf_sums_errors_fwd", "nudf_sums_errors_bwd", "nudf_color_loss_fwd", "nudf_color_loss_bwd",
    "nudf_gen_ray_batch", "nudf_color_loss_sums", "nudf_color_loss_finish",
    "nudf_step_loss_fwd", "nudf_step_loss_bwd", "nudf_col0_seed4",
    "nudf_blend_loss_prepare", "nudf_blend_loss_fwd", "nudf_blend_loss_bwd",
]

_P, _I, _F = C.c_void_p, C.c_int, C.c_float
_ARGTYPES = {
    "nudf_gemm_nn": [C.POINTER(GemmNN), _P],
    "nudf_gemm_tn": [C.POINTER(GemmTN), _P],
    "nudf_gemm_tn_grouped": [C.POINTER(GemmTNGroup), _P],
    "nudf_gemm_tn_grouped_workspace": [C.POINTER(GemmTNGroup)],
    "nudf_gemm_tn_grouped_plan": [C.POINTER(GemmTNGroup), C.POINTER(C.c_int32), i32],
    "nudf_set_tn_flags": [i32],
    "nudf_set_tn_debug": [_P],
    "nudf_composite_fwd": [C.POINTER(Composite), _P],
    "nudf_composite_bwd": [C.POINTER(Composite), C.POINTER(CompositeGrad), _P],
    "nudf_partial_sums": [_P, _I, _I, _P, _P],
    "nudf_composite_colour_finish": [_P, _P, _I, _I, _P, _P, _P, _P, _P],
    "nudf_upsample": [C.POINTER(Upsample), _P],
    "nudf_merge": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "nudf_coarse_z": [_P, _P, _I, _P, _I, _I, _P, _P, _P],
    "nudf_coarse_start": [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "nudf_merge_points": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "nudf_outside_z": [_P, _I, _P, _I, _I, _I, _P, _P],
    "nudf_ray_points": [_P, _P, _P, _P, _I, _I, _I, _P, _P],
    "nudf_posenc": [_P, _I, _I, _P, _I, _I, _F, _I, _P, _I, _F, _P, _I, _F, _P],
    "nudf_posenc_vjp": [_P, _I, _I, _I, _F, _I, _P, _I, _F, _P, _I, _F, _P, _P],
    "nudf_copy_cols": [_P, _I, _I, _P, _I, _I, _I, _F, _P],
    "nudf_add_cols": [_P, _I, _P, _I, _P, _I, _I, _I, _P],
    "nudf_udf_grad_seed": [_P, _P, _P, _I, _F, _I, _I, _F, _P, _I, _P],
    "nudf_udf_head_bwd": [_P, _P, _P, _I, _I, _I, _F, _P, _I, _P],
    "nudf_signed_colsum": [_P, _P, _I, _I, _I, _F, _P, _P],
    "nudf_sigmoid_head_bwd": [_P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P],
    "nudf_weightnorm_pack": [_P, _P, _I, _I, _P, _P, _I, _P, _I, _P, _P],
    "nudf_weightnorm_unpack_grad": [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "nudf_pixel_blend_fwd": [C.POINTER(PixelBlend), _P],
    "nudf_pixel_blend_bwd": [C.POINTER(PixelBlend), _P, _P, _P],
    "nudf_pixel_composite_fwd": [C.POINTER(PixelComposite), _P],
    "nudf_pixel_composite_bwd": [C.POINTER(PixelComposite), _P, _P, _P, _P, _P, _P],
    "nudf_patch_blend_fwd": [C.POINTER(PatchBlend), _P],
    "nudf_patch_blend_bwd": [C.POINTER(PatchBlend), _P, _P, _P, _P],
    "nudf_pixel_warp": [C.POINTER(PixelBlend), _P, _P, _P],
    "nudf_patch_warp": [C.POINTER(PatchWarp), _P],
    "nudf_ssim_patch": [_P, _P, _P, _I, _I, _P, _P, _P, _P],
    "nudf_patch_metric": [_I] + [_P, _P, _P, _I, _I, _P, _P, _P, _P],
    "nudf_adam_step": [C.POINTER(Adam), _P],
    "nudf_mlp_chain": [C.POINTER(Chain), _P],
    "nudf_pack_frag": [_P, _I, _I, _I, _P, _P],
    "nudf_weightnorm_pack_multi": [C.POINTER(PackMulti), _P],
    "nudf_weightnorm_unpack_grad_multi": [C.POINTER(UnpackMulti), _P],
    "nudf_scalars_fwd": [_P, _P, _P, _F, _P, _P, _P],
    "nudf_scalars_bwd": [_P, _P, _P, _F, _P, _P, _P],
    "nudf_l1_sum_fwd": [_P, _P, _I, _P, _P],
    "nudf_l1_sum_bwd": [_P, _P, _I, _P, _P, _P],
    "nudf_sums_errors_fwd": [_P, _F, _P, _P],
    "nudf_sums_errors_bwd": [_P, _F, _P, _P, _P],
    "nudf_color_loss_fwd": [_P, _P, _P, _I, _P, _I, _F, _F, _F, _P, _P, _P, _P],
    "nudf_color_loss_bwd": [_P, _P, _P, _I, _P, _F, _F, _F, _P, _P, _P, _P, _P],
    "nudf_gen_ray_batch": [C.POINTER(RayBatch), _P],
    "nudf_color_loss_sums": [_P, _P, _P, _I, _P, _I, _P, _P],
    "nudf_color_loss_finish": [_P, _I, _F, _F, _F, _P, _P, _P, _P],
    "nudf_step_loss_fwd": [_P, _P, _P, _I, _P, _I, _P, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _I, _P],
    "nudf_step_loss_bwd": [_P, _P, _P, _I, _P, _P, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "nudf_col0_seed4": [_P, _P, _F, _I, _I, _P, _P, _P],
    "nudf_blend_loss_prepare": [C.POINTER(BlendLoss), _P],
    "nudf_blend_loss_fwd": [C.POINTER(BlendLoss), _P],
    "nudf_blend_loss_bwd": [C.POINTER(BlendLoss), _P],
}

_lib = None


def _bind(l):
    for name, at in _ARGTYPES.items():
        fn = getattr(l, name)
        fn.argtypes = at
        fn.restype = C.c_int


def lib():
    """The loaded library; raises NudfError when it has not been built (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NudfError(
                f"{LIB_PATH} not found: build it with `python -m neuraludf_amd.build` "
                "(the NeuralUDF hot path has no CPU/PyTorch fallback)")
        try:
            _lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NudfError(f"cannot load {LIB_PATH}: {e}") from e
        _lib.nudf_last_error.restype = C.c_char_p
        _lib.nudf_version.restype = C.c_int
        _lib.nudf_adam_chunk.restype = C.c_int
        if _lib.nudf_version() != ABI_VERSION:      # the ctypes structures below mirror include/nudf.h of exactly this version
            v, _lib = _lib.nudf_version(), None
            raise NudfError(f"{LIB_PATH} is ABI version {v}, this package binds {ABI_VERSION}: rebuild with "
                            "`python -m neuraludf_amd.build --force`")
        _bind(_lib)
        _lib.nudf_gemm_tn_grouped_workspace.restype = C.c_int64
        _lib.nudf_set_chain_t16.argtypes = [C.c_int]
        _lib.nudf_set_chain_t16.restype = C.c_int
        _lib.nudf_set_status_flag.argtypes = [C.c_void_p]
        _lib.nudf_set_status_flag.restype = C.c_int
        _lib.nudf_status_flag.argtypes = []
        _lib.nudf_status_flag.restype = C.c_void_p
    return _lib


# ---- the non-finite status word (include/nudf.h: nudf_set_status_flag) -----------------------------------------------
STATUS_NONFINITE_RENDER, STATUS_NONFINITE_SAMPLES, STATUS_NONFINITE_LOSS = 1, 2, 4
_STATUS_WORDS = {}        # device index -> the int32 tensor the kernels OR their bits into
_STATUS_BOUND = None      # device index the library currently points at


def bind_status(dev):
    """make the kernels' status word the one of `dev` (allocated and zeroed on first use).  One python compare per call
    once bound; the library keeps one pointer per process (one process per GPU)."""
    global _STATUS_BOUND
    if dev.type != "cuda":
        raise NudfError("libnudf kernels need device tensors (got a CPU tensor): run on an MI355X")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if _STATUS_BOUND == idx:
        return _STATUS_WORDS[idx]
    w = _STATUS_WORDS.get(idx)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
        _STATUS_WORDS[idx] = w
    check(lib().nudf_set_status_flag(C.c_void_p(w.data_ptr())), "nudf_set_status_flag")
    _STATUS_BOUND = idx
    return w


def read_status(dev, clear=False):
    """the status bits seen so far on `dev` (synchronises: one 4-byte D2H copy)."""
    w = bind_status(dev)
    v = int(w.item())
    if clear and v:
        w.zero_()
    return v


def status_text(bits):
    names = [(STATUS_NONFINITE_SAMPLES, "non-finite new samples in the hierarchical re-sampling (nudf_upsample)"),
             (STATUS_NONFINITE_RENDER, "non-finite composited ray outputs or renderer scalars (nudf_composite_fwd)"),
             (STATUS_NONFINITE_LOSS, "non-finite loss (nudf_step_loss_fwd)")]
    return "; ".join(t for b, t in names if bits & b) or "finite"


def ptr(t):
    """raw device pointer of a contiguous fp32 / int32 / bf16 CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NudfError("libnudf kernels need device tensors (got a CPU tensor): run on an MI355X")
    if not t.is_contiguous():
        raise NudfError("libnudf kernels need contiguous tensors")
    return t.data_ptr()


HOST_FAST = os.environ.get("NUDF_HOST_FAST", "1") != "0"      # 0: the round-5 host path (A/B of the eager host cost)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """torch's current stream of the current device as a raw hipStream_t.  (torch.cuda.current_stream() builds a Stream object
    through three layers of Python argument checking: 12 us per call, 19 launches per forward pass -- the raw accessors of
    the same state cost 0.3 us.)"""
    if HOST_FAST and _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what=""):
    if rc != 0:
        msg = lib().nudf_last_error()
        raise NudfError(f"{what} failed (hipError {rc}): {msg.decode() if msg else ''}")


_FN = {}


def call(name, *args):
    """call an entry point on torch's current stream; struct arguments are passed by reference."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    rc = fn(*[C.byref(a) if isinstance(a, C.Structure) else a for a in args], stream())
    if rc != 0:
        check(rc, name)
