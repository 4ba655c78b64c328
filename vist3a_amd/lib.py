"""ctypes binding of libvist3a_hip.so (C ABI declared in include/vist3a_hip.h).

The product path has no fallback: if the shared object is missing or fails to load, every op raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("V3A_LIB") or _HERE / "libvist3a_hip.so")   # V3A_LIB: an experiment build of the same ABI (tools/variant_build.sh)

V3A_OK = 0
ERRORS = {-1: "V3A_ERR_ARG", -2: "V3A_ERR_SHAPE", -3: "V3A_ERR_LAUNCH", -4: "V3A_ERR_WORKSPACE"}

ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU, ACT_RELU = 0, 1, 2, 3, 4
GEMM_BIAS_ROW = 1 << 0
GEMM_SCALE_PER_BATCH = 1 << 1
GEMM_ROUND_AFTER_SCALE = 1 << 2
GEMM_RES_F32 = 1 << 3
GEMM_OUT_F32 = 1 << 4
GEMM_RELU_OUT = 1 << 6


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("scale", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int),
        ("rows_per_batch", C.c_int), ("scale_stride", C.c_int),
        ("act", C.c_int), ("flags", C.c_int), ("tile", C.c_int),
        ("residual2", C.c_void_p), ("ldr2", C.c_int), ("res_row_mod", C.c_int),
        ("out_row_group", C.c_int), ("out_row_skip", C.c_int), ("out_row_off", C.c_int),
        ("split_k", C.c_int), ("workspace", C.c_void_p),
        ("batch", C.c_int), ("a_batch_stride", C.c_long), ("b_batch_stride", C.c_long), ("c_batch_stride", C.c_long),
        ("res_batch_stride", C.c_long), ("row_sumsq", C.c_void_p),
        ("C_t", C.c_void_p), ("ldct", C.c_int), ("t_col0", C.c_int),
    ]


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("ktab", C.c_void_p), ("y", C.c_void_p),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("scale", C.c_void_p),
        ("T", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int),
        ("oT", C.c_int), ("oH", C.c_int), ("oW", C.c_int), ("Cout", C.c_int), ("Kpad", C.c_int),
        ("sT", C.c_int), ("sH", C.c_int), ("sW", C.c_int),
        ("pT", C.c_int), ("pH", C.c_int), ("pW", C.c_int),
        ("ups2", C.c_int), ("replicate", C.c_int),
        ("ldy", C.c_int), ("ldr", C.c_int),
        ("act", C.c_int), ("flags", C.c_int), ("tile", C.c_int),
        ("residual2", C.c_void_p), ("ldr2", C.c_int), ("res_row_mod", C.c_int),
        ("out_row_group", C.c_int), ("out_row_skip", C.c_int), ("out_row_off", C.c_int),
        ("w_halo", C.c_void_p), ("halo_kT", C.c_int),
    ]


class ConvSplitArgs(C.Structure):
    _fields_ = [("c", ConvArgs), ("x_lo", C.c_void_p), ("y_lo", C.c_void_p), ("residual_lo", C.c_void_p), ("residual2_lo", C.c_void_p)]


class GemmSkinnyArgs(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ldx", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int),
        ("act", C.c_int), ("flags", C.c_int), ("transposed_out", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_long),
    ]


class GsProjectArgs(C.Structure):
    _fields_ = [
        ("means", C.c_void_p), ("covars", C.c_void_p), ("sh", C.c_void_p),
        ("sh_layout", C.c_int), ("sh_k", C.c_int), ("sh_degree", C.c_int),
        ("viewmat", C.c_void_p), ("campos", C.c_void_p), ("K", C.c_void_p),
        ("U", C.c_long), ("C", C.c_int), ("width", C.c_int), ("height", C.c_int),
        ("near_plane", C.c_float), ("far_plane", C.c_float), ("radius_clip", C.c_float), ("eps2d", C.c_float),
        ("radii", C.c_void_p), ("means2d", C.c_void_p), ("depths", C.c_void_p), ("conics", C.c_void_p), ("colors", C.c_void_p),
    ]


class UniPCStepArgs(C.Structure):
    _fields_ = [
        ("dit_out", C.c_void_p), ("tok", C.c_void_p), ("sample", C.c_void_p), ("last_sample", C.c_void_p), ("m_prev1", C.c_void_p),
        ("m_prev2", C.c_void_p), ("m_out", C.c_void_p), ("sample_corrected", C.c_void_p), ("prev", C.c_void_p),
        ("C", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int), ("batch", C.c_int), ("guided", C.c_int),
        ("guidance", C.c_float), ("sigma", C.c_float),
        ("corr_order", C.c_int), ("cc1", C.c_float), ("cc2", C.c_float), ("cc3", C.c_float), ("c_rho_last", C.c_float), ("c_rho0", C.c_float),
        ("c_inv_rk", C.c_float),
        ("pred_order", C.c_int), ("pc1", C.c_float), ("pc2", C.c_float), ("pc3", C.c_float), ("p_rho0", C.c_float), ("p_inv_rk", C.c_float),
    ]


class GsRasterizeArgs(C.Structure):
    _fields_ = [
        ("radii", C.c_void_p), ("means2d", C.c_void_p), ("depths", C.c_void_p), ("conics", C.c_void_p), ("colors", C.c_void_p),
        ("opacities", C.c_void_p), ("background", C.c_void_p),
        ("U", C.c_long), ("C", C.c_int), ("width", C.c_int), ("height", C.c_int), ("clamp_rgb", C.c_int),
        ("out_color", C.c_void_p), ("out_depth", C.c_void_p), ("out_alpha", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_long), ("max_isect", C.c_long),
        ("n_isect", C.POINTER(C.c_long)),
        ("tile_offsets_out", C.c_void_p), ("flatten_ids_out", C.c_void_p),
    ]



class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("o", C.c_void_p),
        ("q_batch_stride", C.c_long), ("k_batch_stride", C.c_long),
        ("vt_batch_stride", C.c_long), ("o_batch_stride", C.c_long),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldvt", C.c_int), ("ldo", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int), ("D", C.c_int),
        ("scale", C.c_float), ("kv_period", C.c_int), ("kv_valid", C.c_int),
        ("rel_bias", C.c_void_p), ("rel_bias_stride", C.c_int), ("rel_bias_center", C.c_int),
        ("key_bias", C.c_void_p), ("key_bias_stride", C.c_int), ("key_bias_first", C.c_int),
        ("kv_seg", C.c_int), ("k_seg_stride", C.c_long), ("vt_seg_stride", C.c_long),
        ("kv_split", C.c_int), ("workspace", C.c_void_p),
    ]


class XattnProbsArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("p", C.c_void_p), ("key_bias", C.c_void_p),
        ("q_batch_stride", C.c_long), ("k_batch_stride", C.c_long), ("p_batch_stride", C.c_long),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldp", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int), ("D", C.c_int), ("Lkp", C.c_int),
        ("key_bias_stride", C.c_int), ("key_bias_first", C.c_int), ("scale", C.c_float),
        ("q_row_sumsq", C.c_void_p), ("q_sumsq_parts", C.c_int), ("q_eps", C.c_float),
    ]


class GemmFp8Args(C.Structure):
    _fields_ = [("g", GemmArgs), ("a_scale", C.c_void_p), ("b_scale", C.c_void_p)]


class AttnFp8Args(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("o", C.c_void_p),
        ("q_batch_stride", C.c_long), ("k_batch_stride", C.c_long),
        ("vt_batch_stride", C.c_long), ("o_batch_stride", C.c_long),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldvt", C.c_int), ("ldo", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int), ("D", C.c_int),
        ("scale", C.c_float), ("q_scale", C.c_float), ("k_scale", C.c_float), ("v_scale", C.c_float),
        ("kv_seg", C.c_int), ("k_seg_stride", C.c_long), ("vt_seg_stride", C.c_long),
        ("kv_split", C.c_int), ("workspace", C.c_void_p),
    ]


class LayerNormArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p),
        ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("M", C.c_int), ("d", C.c_int), ("ldx", C.c_int), ("ldy", C.c_int),
        ("rows_per_batch", C.c_int), ("mod_stride", C.c_int),
        ("eps", C.c_float),
        ("x_is_f32", C.c_int), ("y_is_f32", C.c_int),
        ("in_row_group", C.c_int), ("in_row_skip", C.c_int), ("in_row_off", C.c_int),
        ("out_row_group", C.c_int), ("out_row_skip", C.c_int), ("out_row_off", C.c_int),
        ("rms", C.c_int),
        ("y_fp8_scale", C.c_void_p),
    ]


class RmsNormRopeArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p),
        ("weight", C.c_void_p), ("rope", C.c_void_p),
        ("M", C.c_int), ("d", C.c_int), ("ldx", C.c_int), ("ldy", C.c_int),
        ("head_dim", C.c_int), ("tokens_per_batch", C.c_int),
        ("eps", C.c_float),
        ("weight2", C.c_void_p),
        ("y_fp8_scale", C.c_float),
    ]


class RowNormArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("M", C.c_long), ("d", C.c_int), ("ldx", C.c_int), ("ldy", C.c_int),
        ("eps", C.c_float), ("mode", C.c_int), ("act", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/vist3a_hip.h declares must be listed here
SYMBOLS = {
    "v3a_abi_version": (C.c_int, []),
    "v3a_build_info": (C.c_char_p, []),
    "v3a_gemm_bf16_nt": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "v3a_gemm_num_tiles": (C.c_int, []),
    "v3a_gemm_pick_tile": (C.c_int, [C.c_int, C.c_int]),
    "v3a_gemm_tile_name": (C.c_char_p, [C.c_int]),
    "v3a_conv_bf16": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "v3a_conv_halo_tiles": (C.c_long, [C.POINTER(ConvArgs)]),
    "v3a_conv_split": (C.c_int, [C.POINTER(ConvSplitArgs), C.c_void_p]),
    "v3a_conv_split_halo_bn": (C.c_int, [C.c_int]),
    "v3a_conv_split_halo_tiles": (C.c_long, [C.POINTER(ConvSplitArgs)]),
    "v3a_split_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "v3a_layernorm_pair": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float] + [C.c_int] * 3 + [C.c_void_p]),
    "v3a_bilinear_cl_pair": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 7 + [C.c_void_p]),
    "v3a_attention_fwd_bf16": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    "v3a_xattn_probs_bf16": (C.c_int, [C.POINTER(XattnProbsArgs), C.c_void_p]),
    "v3a_attention_split_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "v3a_unipc_cfg_step": (C.c_int, [C.POINTER(UniPCStepArgs), C.c_void_p]),
    "v3a_attention_fwd_fp8": (C.c_int, [C.POINTER(AttnFp8Args), C.c_void_p]),
    "v3a_gemm_split_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "v3a_gemm_pick_tile_act": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "v3a_gemm_pick_tile_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "v3a_gemm_fp8_nt": (C.c_int, [C.POINTER(GemmFp8Args), C.c_void_p]),
    "v3a_gemm_fp8_num_tiles": (C.c_int, []),
    "v3a_gemm_fp8_pick_tile": (C.c_int, [C.c_int, C.c_int]),
    "v3a_gemm_fp8_tile_name": (C.c_char_p, [C.c_int]),
    "v3a_quantize_fp8_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "v3a_quantize_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "v3a_layernorm": (C.c_int, [C.POINTER(LayerNormArgs), C.c_void_p]),
    "v3a_rmsnorm_rope": (C.c_int, [C.POINTER(RmsNormRopeArgs), C.c_void_p]),
    "v3a_rownorm_act": (C.c_int, [C.POINTER(RowNormArgs), C.c_void_p]),
    "v3a_qknorm_rope2d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "v3a_latent_upsample_t_cl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "v3a_bilinear_cl": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_void_p]),
    "v3a_depth_unproject": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "v3a_voxelize_workspace_bytes": (C.c_long, [C.c_long]),
    "v3a_voxelize_fuse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_float, C.c_void_p, C.c_long]
                          + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v3a_conf_compact_workspace_bytes": (C.c_long, [C.c_long]),
    "v3a_conf_quantile_compact": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_long,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "v3a_gaussian_adapter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_float] + [C.c_void_p] * 8),
    "v3a_linear_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]),
    "v3a_attention_small_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "v3a_gemm_skinny_workspace_bytes": (C.c_long, [C.c_int, C.c_int, C.c_int]),
    "v3a_gemm_skinny_bf16": (C.c_int, [C.POINTER(GemmSkinnyArgs), C.c_void_p]),
    "v3a_gs_project": (C.c_int, [C.POINTER(GsProjectArgs), C.c_void_p]),
    "v3a_gs_rasterize_workspace_bytes": (C.c_long, [C.c_long, C.c_int, C.c_int, C.c_int, C.c_long]),
    "v3a_gs_rasterize": (C.c_int, [C.POINTER(GsRasterizeArgs), C.c_void_p]),
    "v3a_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
}

_lib = None
EXPECTED_ABI = 21   # = v3a_abi_version() of csrc/capi.hip; bumped together with every struct / signature change in include/vist3a_hip.h


class HipLibraryError(RuntimeError):
    pass


def load(path: os.PathLike | None = None) -> C.CDLL:
    """Load (once) and type the shared library.  Raises HipLibraryError when it is absent — there is no
    CPU or eager-PyTorch fallback on the product path."""
    global _lib
    if _lib is not None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise HipLibraryError(
            f"{p} not found: build it with `python -m vist3a_amd.build` (hipcc, gfx950). "
            "The VIST3A MI355X path has no fallback implementation."
        )
    # The library links against libamdhip64; PyTorch-ROCm ships its own copy.  Whichever is mapped FIRST becomes the process's HIP runtime
    # for that soname: if this .so came first, its kernels would be registered with /opt/rocm's runtime while every stream and device
    # pointer it is handed belongs to torch's - the first launch then fails with V3A_ERR_LAUNCH (seen when build() and smoke() ran in one
    # process).  Import torch first so that there is exactly one runtime.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(str(p))
    except OSError as e:  # pragma: no cover - depends on the ROCm runtime being present
        raise HipLibraryError(f"failed to load {p}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.v3a_abi_version()
    if got != EXPECTED_ABI:   # a stale in-tree .so would read the structs above past their old end
        raise HipLibraryError(f"{p} has ABI version {got}, this package binds version {EXPECTED_ABI}: rebuild it with "
                              "`python -m vist3a_amd.build`")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != V3A_OK:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")
