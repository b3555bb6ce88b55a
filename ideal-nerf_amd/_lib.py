"""ctypes binding of libidealnerf.so (include/idealnerf.h).

No torch types cross this boundary: only raw device addresses, sizes and the HIP
stream handle.  The library is built in-tree by ``ideal-nerf_amd/build.py``
(``__graft_entry__.build()``); if it is missing the import of any op fails loudly --
there is no CPU or eager-PyTorch fallback in the product path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDN_LIB") or os.path.join(HERE, "libidealnerf.so")

IDN_PREC_F32, IDN_PREC_BF16X3, IDN_PREC_BF16, IDN_PREC_FP16X3, IDN_PREC_BF16X6 = 0, 1, 2, 3, 4
RAY_FLOATS = 11
PROF_KINDS = ("mlp_fwd", "mlp_fwd_save", "delta_chain", "dw_gemm", "dw_gemm_x6", "mlp_fwd_save_x6", "delta_chain_x6")   # IDN_PROF_* of include/idealnerf.h

fp = C.c_void_p  # device pointers travel as integers


class FaceNerfParams(C.Structure):
    _fields_ = [("pts_w", fp * 8), ("pts_b", fp * 8), ("views_w", fp * 3), ("views_b", fp * 3),
                ("alpha_w", fp), ("alpha_b", fp), ("rgb_w", fp), ("rgb_b", fp),
                ("dim_aud", C.c_int), ("dim_expr", C.c_int), ("dim_latent", C.c_int)]


class FaceNerfGrads(C.Structure):
    _fields_ = [("pts_w", fp * 8), ("pts_b", fp * 8), ("views_w", fp * 3), ("views_b", fp * 3),
                ("alpha_w", fp), ("alpha_b", fp), ("rgb_w", fp), ("rgb_b", fp)]


class CompositeOut(C.Structure):
    _fields_ = [(n, fp) for n in ("rgb_map", "disp_map", "acc_map", "depth_map", "weights", "rgb_fg", "last_weight")]


class RenderArgs(C.Structure):
    _fields_ = [("rays", fp), ("bc_rgb", fp), ("n_rays", C.c_int64), ("n_samples", C.c_int),
                ("n_importance", C.c_int), ("precision", C.c_int),
                ("packed_coarse", fp), ("folded_coarse", fp), ("packed_fine", fp), ("folded_fine", fp),
                ("t_vals", fp), ("t_rand", fp), ("u", fp), ("u_per_ray", C.c_int)] + \
               [(n, fp) for n in ("rgb_map", "disp_map", "acc_map", "depth_map", "last_weight", "rgb_fg",
                                  "rgb0", "disp0", "acc0", "z_std", "last_weight0", "rgb_fg0",
                                  "tap_z_coarse", "tap_raw_coarse", "tap_weights_coarse", "tap_cdf", "tap_inds",
                                  "tap_z_samples", "tap_z_fine", "tap_raw_fine", "tap_weights_fine")] + \
               [("workspace", fp), ("workspace_bytes", C.c_size_t), ("precision_fine_plus1", C.c_int),
                ("lindisp", C.c_int), ("white_bkgd", C.c_int), ("noise_coarse", fp), ("noise_fine", fp), ("fused_march", C.c_int),
                ("rng_mode", C.c_int), ("rng_seed", C.c_uint64), ("rng_ray0", C.c_int64)]


class AudioNetParams(C.Structure):
    _fields_ = [("conv_w", fp * 4), ("conv_b", fp * 4), ("fc_w", fp * 2), ("fc_b", fp * 2), ("dim_aud", C.c_int)]


class AudioNetGrads(C.Structure):
    _fields_ = [("conv_w", fp * 4), ("conv_b", fp * 4), ("fc_w", fp * 2), ("fc_b", fp * 2)]


class Frame(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("H", C.c_int), ("W", C.c_int), ("focal", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("near_", C.c_float), ("far_", C.c_float), ("row0", C.c_int), ("nrows", C.c_int), ("rays_out", fp)]


ABI_VERSION = 4   # idealnerf_version(): 4 since idn_render_args carries `rng_mode / rng_seed / rng_ray0` (3: `fused_march`)

# name -> (restype, argtypes); mirrors include/idealnerf.h one to one
PROTOTYPES = {
    "idealnerf_version": (C.c_int, []),
    "idealnerf_last_error": (C.c_char_p, []),
    "idealnerf_packed_weight_floats": (C.c_size_t, [C.c_int]),
    "idealnerf_folded_bias_floats": (C.c_size_t, []),
    "idealnerf_pack_weights": (C.c_int, [C.POINTER(FaceNerfParams), C.c_int, fp, fp]),
    "idealnerf_fold_conditioning": (C.c_int, [C.POINTER(FaceNerfParams), fp, fp, fp, fp, fp]),
    "idealnerf_facenerf_fwd": (C.c_int, [fp, fp, C.c_int, fp, C.c_int64, fp, fp]),
    "idealnerf_query_rays_fwd": (C.c_int, [fp, fp, C.c_int, fp, fp, C.c_int64, C.c_int, fp, fp]),
    "idealnerf_query_points_fwd": (C.c_int, [fp, fp, C.c_int, fp, fp, C.c_int64, C.c_int, fp, fp]),
    "idealnerf_frame_rays": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_int, C.c_int, fp, fp]),
    "idealnerf_to8b": (C.c_int, [fp, C.c_int64, C.c_int, fp, fp, fp]),
    "idealnerf_philox_uniform": (C.c_int, [C.c_uint64, C.c_int, C.c_int64, C.c_int64, C.c_int, fp, fp]),
    "idealnerf_coarse_depths": (C.c_int, [fp, fp, fp, C.c_int, C.c_int64, C.c_int, fp, fp]),
    "idealnerf_composite_fwd": (C.c_int, [fp, fp, fp, fp, fp, C.c_int, C.c_int64, C.c_int, C.POINTER(CompositeOut), fp]),
    "idealnerf_sample_pdf_fwd": (C.c_int, [fp, fp, fp, C.c_int, C.c_int64, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp]),
    "idealnerf_march_fwd": (C.c_int, [fp, fp, fp, fp, fp, C.c_int, fp, C.c_int, C.c_int64, C.c_int, C.c_int,
                                      C.POINTER(CompositeOut), fp, fp, fp, fp, fp, fp]),
    "idealnerf_sample_pdf_bins_fwd": (C.c_int, [fp, fp, fp, C.c_int, C.c_int64, C.c_int, C.c_int, fp, fp, fp, fp]),
    "idealnerf_invert_cdf": (C.c_int, [fp, fp, fp, C.c_int, C.c_int64, C.c_int, C.c_int, fp, fp, fp]),
    "idealnerf_render_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "idealnerf_render_rays_fwd": (C.c_int, [C.POINTER(RenderArgs), fp]),
    "idealnerf_render_frame_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "idealnerf_render_frame_fwd": (C.c_int, [C.POINTER(RenderArgs), C.POINTER(Frame), fp]),
    "idealnerf_audio_net_saved_floats": (C.c_size_t, [C.c_int]),
    "idealnerf_audio_net_fwd": (C.c_int, [C.POINTER(AudioNetParams), fp, C.c_int, fp, fp, fp]),
    "idealnerf_audio_net_bwd": (C.c_int, [C.POINTER(AudioNetParams), C.POINTER(AudioNetGrads), fp, fp, fp, C.c_int, fp]),
    "idealnerf_train_acts_floats": (C.c_size_t, [C.c_int64]),
    "idealnerf_query_rays_train_fwd": (C.c_int, [fp, fp, C.c_int, fp, fp, C.c_int64, C.c_int, fp, fp, fp]),
    "idealnerf_pass_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "idealnerf_pass_bwd": (C.c_int, [C.POINTER(FaceNerfParams), C.POINTER(FaceNerfGrads), fp, fp, fp, fp, fp, fp, fp,
                                     fp, C.c_int64, C.c_int, fp, fp, fp, fp, fp, fp, fp, C.c_size_t, fp]),
    "idealnerf_dw_gemm_workspace_bytes": (C.c_size_t, []),
    "idealnerf_dw_gemm": (C.c_int, [fp, C.c_int, fp, C.c_int, C.c_int64, fp, fp, C.c_int, fp, C.c_size_t, fp]),
    "idealnerf_profile_begin": (None, []),
    "idealnerf_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "idealnerf_profile_end_kinds": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_lib = None


class IdealNerfError(RuntimeError):
    pass


def load():
    """Load the shared library once and attach prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IdealNerfError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python ideal-nerf_amd/build.py, or __graft_entry__.build()). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    if lib.idealnerf_version() != ABI_VERSION:   # struct layouts above belong to one ABI version (IDN_LIB may point at an older build)
        raise IdealNerfError(f"{LIB_PATH} has ABI version {lib.idealnerf_version()}, this package binds version {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().idealnerf_last_error()
        raise IdealNerfError(f"libidealnerf error {rc}: {msg.decode() if msg else '?'}")
