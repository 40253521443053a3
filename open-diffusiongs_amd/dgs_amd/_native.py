"""ctypes binding of the C ABI declared in include/dgs_raster.h (and include/dgs_dit.h).

The product path loads exactly one library: open-diffusiongs_amd/lib/libdgs_hip.so (hipcc, gfx950).
If it is missing the import fails loudly -- there is NO CPU / PyTorch fallback.  (The test-suite's CPU
emulation build is opened explicitly by tests through `open_library(path)`; nothing here looks for it.)
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DGS_AMD_LIBRARY: another build of the SAME sources -- the instrumented library of tools/ (`DGS_INSTRUMENT=1 python -m dgs_amd.build`)
# or an A/B side (tools/ab_build.sh).  Never a fallback: whatever is named must exist and pass the ABI check.
LIB_PATH = os.environ.get("DGS_AMD_LIBRARY") or os.path.join(os.path.dirname(HERE), "lib", "libdgs_hip.so")

ABI_VERSION = 8          # == DGS_ABI_VERSION of include/dgs_raster.h; bump both whenever a struct or prototype changes
DGS_ERR_BINNING_OVERFLOW = -7    # include/dgs_raster.h DgsStatus
c_float_p = ctypes.POINTER(ctypes.c_float)
ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class DgsRasterForwardArgs(ctypes.Structure):
    _fields_ = [
        ("P", ctypes.c_int32), ("D", ctypes.c_int32), ("M", ctypes.c_int32),
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("V", ctypes.c_int32), ("views_per_set", ctypes.c_int32),
        ("background", ctypes.c_void_p), ("means3D", ctypes.c_void_p), ("shs", ctypes.c_void_p),
        ("colors_precomp", ctypes.c_void_p), ("opacities", ctypes.c_void_p), ("scales", ctypes.c_void_p),
        ("rotations", ctypes.c_void_p), ("cov3D_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
        ("projmatrix", ctypes.c_void_p), ("campos", ctypes.c_void_p), ("tanfov", ctypes.c_void_p),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float),
        ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32), ("raw_activations", ctypes.c_int32),
        ("out_color", ctypes.c_void_p), ("radii", ctypes.c_void_p),
        ("geom_alloc", ALLOC_FN), ("geom_user", ctypes.c_void_p),
        ("img_alloc", ALLOC_FN), ("img_user", ctypes.c_void_p),
        ("binning_alloc", ALLOC_FN), ("binning_user", ctypes.c_void_p),
        ("binning_capacity", ctypes.c_int64), ("num_rendered_dev", ctypes.c_void_p), ("num_rendered_host", ctypes.c_void_p), ("longest_hint", ctypes.c_int64),
        ("num_rendered", ctypes.c_int64), ("longest_list", ctypes.c_int64), ("binning_form", ctypes.c_int32), ("exact_exp", ctypes.c_int32),
    ]


class DgsRasterBackwardArgs(ctypes.Structure):
    _fields_ = [
        ("P", ctypes.c_int32), ("D", ctypes.c_int32), ("M", ctypes.c_int32), ("width", ctypes.c_int32),
        ("height", ctypes.c_int32), ("V", ctypes.c_int32), ("views_per_set", ctypes.c_int32),
        ("num_rendered", ctypes.c_int64),
        ("background", ctypes.c_void_p), ("means3D", ctypes.c_void_p), ("shs", ctypes.c_void_p),
        ("colors_precomp", ctypes.c_void_p), ("opacities", ctypes.c_void_p), ("scales", ctypes.c_void_p),
        ("rotations", ctypes.c_void_p), ("cov3D_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
        ("projmatrix", ctypes.c_void_p), ("campos", ctypes.c_void_p), ("tanfov", ctypes.c_void_p),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float),
        ("debug", ctypes.c_int32), ("raw_activations", ctypes.c_int32),
        ("radii", ctypes.c_void_p), ("dL_dpix", ctypes.c_void_p),
        ("geom_buffer", ctypes.c_void_p), ("binning_buffer", ctypes.c_void_p), ("img_buffer", ctypes.c_void_p),
        ("dL_dmeans2D", ctypes.c_void_p), ("dL_dconic", ctypes.c_void_p), ("dL_dcolors", ctypes.c_void_p),
        ("dL_dcov3D", ctypes.c_void_p), ("dL_dopacity", ctypes.c_void_p), ("dL_dmeans3D", ctypes.c_void_p),
        ("dL_dsh", ctypes.c_void_p), ("dL_dscales", ctypes.c_void_p), ("dL_drotations", ctypes.c_void_p),
        ("exact_exp", ctypes.c_int32), ("scratch", ctypes.c_void_p), ("scratch_bytes", ctypes.c_size_t),
    ]


# every symbol include/dgs_raster.h declares (checked by tests/test_abi.py)
RASTER_SYMBOLS = ["dgs_abi_version", "dgs_status_string", "dgs_raster_geom_bytes", "dgs_raster_image_bytes",
                  "dgs_raster_binning_bytes", "dgs_raster_forward", "dgs_raster_binning_form", "dgs_raster_backward",
                  "dgs_raster_backward_scratch_bytes", "dgs_mark_visible",
                  "dgs_raster_state_read", "dgs_cameras_from_c2w", "dgs_rays_from_c2w"]


def _declare(L):
    L.dgs_abi_version.restype = ctypes.c_int
    L.dgs_status_string.restype = ctypes.c_char_p
    L.dgs_status_string.argtypes = [ctypes.c_int]
    L.dgs_raster_geom_bytes.restype = ctypes.c_size_t
    L.dgs_raster_geom_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
    L.dgs_raster_image_bytes.restype = ctypes.c_size_t
    L.dgs_raster_image_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    L.dgs_raster_binning_bytes.restype = ctypes.c_size_t
    L.dgs_raster_binning_bytes.argtypes = [ctypes.c_int64]
    L.dgs_raster_forward.restype = ctypes.c_int
    L.dgs_raster_forward.argtypes = [ctypes.POINTER(DgsRasterForwardArgs), ctypes.c_void_p]
    L.dgs_raster_binning_form.restype = ctypes.c_int
    L.dgs_raster_binning_form.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    if hasattr(L, "dgs_raster_backward"):
        L.dgs_raster_backward.restype = ctypes.c_int
        L.dgs_raster_backward.argtypes = [ctypes.POINTER(DgsRasterBackwardArgs), ctypes.c_void_p]
        L.dgs_raster_backward_scratch_bytes.restype = ctypes.c_size_t
        L.dgs_raster_backward_scratch_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]
    L.dgs_mark_visible.restype = ctypes.c_int
    L.dgs_mark_visible.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p]
    L.dgs_cameras_from_c2w.restype = ctypes.c_int
    L.dgs_cameras_from_c2w.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]
    L.dgs_rays_from_c2w.restype = ctypes.c_int
    L.dgs_rays_from_c2w.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.dgs_raster_state_read.restype = ctypes.c_int64
    L.dgs_raster_state_read.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_void_p]
    return L


def open_library(path):
    """Open a library exporting the dgs C ABI and attach prototypes.  Whatever build it is (product, tools' instrumented build, CPU
    emulation), it has to speak THIS binding's struct layouts: a stale build is refused, not called with misaligned structs."""
    L = ctypes.CDLL(path)
    L.dgs_abi_version.restype = ctypes.c_int
    if L.dgs_abi_version() != ABI_VERSION:
        raise RuntimeError(f"dgs_amd: ABI version mismatch: {path} reports {L.dgs_abi_version()}, this binding is written for {ABI_VERSION} "
                           f"(include/dgs_raster.h DGS_ABI_VERSION) -- rebuild it (`python -m dgs_amd.build`, `DGS_INSTRUMENT=1 python -m dgs_amd.build`, "
                           f"tests/hipemu/build_emu.py)")
    return _declare(L)


_lib = None


def lib():
    """The product library (HIP, gfx950).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"dgs_amd: HIP library {LIB_PATH} is missing. Build it with `python -m dgs_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm ships its own copy of the HIP runtime (torch/lib/libamdhip64.so); the device pointers and streams this
        # library is handed come from it.  It has to be in the process FIRST, so that libdgs_hip.so's libamdhip64.so.7 resolves to
        # that copy: loaded the other way round the process holds two runtimes and every launch fails with hipErrorNoDevice.
        import torch  # noqa: F401
        _lib = open_library(LIB_PATH)
        if _lib.dgs_abi_version() != ABI_VERSION:
            raise RuntimeError(f"dgs_amd: ABI version mismatch: libdgs_hip.so reports {_lib.dgs_abi_version()}, this binding is written "
                               f"for {ABI_VERSION} (include/dgs_raster.h DGS_ABI_VERSION) -- rebuild with `python -m dgs_amd.build`")
    return _lib


def status_string(L, code):
    return L.dgs_status_string(int(code)).decode()


# ---------------------------------------------------------------------------------------------------
# include/dgs_dit.h
# ---------------------------------------------------------------------------------------------------
EPI_BF16, EPI_GELU_BF16, EPI_GATE_RESIDUAL, EPI_F32, EPI_QKV, EPI_DGELU_BF16 = range(6)
GEMM_AUTO, GEMM_SIMPLE128, GEMM_SLICED, GEMM_QUAD, GEMM_SLICED128 = 0, 1, 4, 5, 6


class DgsDitGemmArgs(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("A", ctypes.c_void_p), ("lda", ctypes.c_int32), ("W", ctypes.c_void_p), ("ldw", ctypes.c_int32),
                ("bias", ctypes.c_void_p), ("epilogue", ctypes.c_int32), ("out", ctypes.c_void_p), ("ldo", ctypes.c_int32),
                ("gate", ctypes.c_void_p), ("gate_stride", ctypes.c_int32), ("rows_per_batch", ctypes.c_int32),
                ("vt", ctypes.c_void_p), ("resid", ctypes.c_void_p), ("aux", ctypes.c_void_p), ("k_per_batch", ctypes.c_int32),
                ("a_batch_stride", ctypes.c_int64), ("w_batch_stride", ctypes.c_int64), ("algo", ctypes.c_int32),
                ("valid_rows", ctypes.c_int32), ("q_scale", ctypes.c_float), ("splitk_ws", ctypes.c_void_p)]


class DgsDitAttentionArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("heads", ctypes.c_int32), ("L", ctypes.c_int32), ("lpad", ctypes.c_int32),
                ("qk", ctypes.c_void_p), ("vt", ctypes.c_void_p), ("out", ctypes.c_void_p), ("scale", ctypes.c_float),
                ("ld_qk", ctypes.c_int32), ("k_offset", ctypes.c_int32), ("vt_batch_stride", ctypes.c_int64),
                ("lse2", ctypes.c_void_p), ("q_prescaled", ctypes.c_int32), ("tail_ws", ctypes.c_void_p),
                ("tail_ws_bytes", ctypes.c_size_t), ("tail_mode", ctypes.c_int32)]


class DgsDitAttentionBackwardArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("heads", ctypes.c_int32), ("L", ctypes.c_int32), ("lpad", ctypes.c_int32),
                ("qkv", ctypes.c_void_p), ("qkvT", ctypes.c_void_p), ("o", ctypes.c_void_p), ("dO", ctypes.c_void_p),
                ("dOT", ctypes.c_void_p), ("lse2", ctypes.c_void_p), ("D", ctypes.c_void_p), ("dqkv", ctypes.c_void_p),
                ("scale", ctypes.c_float), ("dqkvT", ctypes.c_void_p), ("bias_part", ctypes.c_void_p)]


class DgsDitLayerNormArgs(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_int32), ("width", ctypes.c_int32), ("x", ctypes.c_void_p), ("weight", ctypes.c_void_p),
                ("shift", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("mod_stride", ctypes.c_int32),
                ("rows_per_batch", ctypes.c_int32), ("eps", ctypes.c_float), ("out", ctypes.c_void_p),
                ("out_f32", ctypes.c_int32)]


class DgsDitRowLinearArgs(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("x", ctypes.c_void_p),
                ("silu_input", ctypes.c_int32), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("silu_output", ctypes.c_int32), ("out", ctypes.c_void_p)]


class DgsDitLayerNormBackwardArgs(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_int32), ("width", ctypes.c_int32), ("x", ctypes.c_void_p), ("dh", ctypes.c_void_p),
                ("dh_f32", ctypes.c_int32), ("weight", ctypes.c_void_p), ("scale", ctypes.c_void_p),
                ("mod_stride", ctypes.c_int32), ("rows_per_batch", ctypes.c_int32), ("eps", ctypes.c_float),
                ("dx_in", ctypes.c_void_p), ("dx_out", ctypes.c_void_p), ("dshift", ctypes.c_void_p),
                ("dscale", ctypes.c_void_p), ("dweight", ctypes.c_void_p), ("scratch", ctypes.c_void_p), ("scratch_bytes", ctypes.c_size_t)]


class DgsDitRowLinearBackwardArgs(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("x", ctypes.c_void_p),
                ("silu_input", ctypes.c_int32), ("W", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dW", ctypes.c_void_p),
                ("db", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("scratch", ctypes.c_void_p), ("scratch_bytes", ctypes.c_size_t)]


class DgsDitGateMulArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("rows", ctypes.c_int32), ("width", ctypes.c_int32), ("dx", ctypes.c_void_p),
                ("y", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("gate_stride", ctypes.c_int32), ("dy", ctypes.c_void_p),
                ("dyT", ctypes.c_void_p), ("dgate", ctypes.c_void_p), ("dbias", ctypes.c_void_p), ("scratch", ctypes.c_void_p),
                ("scratch_bytes", ctypes.c_size_t)]


class DgsDitLayerWeights(ctypes.Structure):
    _fields_ = [("qkv_w", ctypes.c_void_p), ("proj_w", ctypes.c_void_p), ("fc1_w", ctypes.c_void_p), ("fc2_w", ctypes.c_void_p),
                ("qkv_b", ctypes.c_void_p), ("proj_b", ctypes.c_void_p), ("fc1_b", ctypes.c_void_p), ("fc2_b", ctypes.c_void_p)]


class DgsDitModel(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("heads", ctypes.c_int32), ("layers", ctypes.c_int32), ("patch", ctypes.c_int32),
                ("in_channels", ctypes.c_int32), ("n_gaussians", ctypes.c_int32), ("gs_channels", ctypes.c_int32),
                ("scene", ctypes.c_int32), ("relative_plk", ctypes.c_int32),
                ("range_near", ctypes.c_float), ("range_far", ctypes.c_float),
                ("t_w0", ctypes.c_void_p), ("t_b0", ctypes.c_void_p), ("t_w1", ctypes.c_void_p), ("t_b1", ctypes.c_void_p),
                ("tok_w", ctypes.c_void_p), ("pos_emb", ctypes.c_void_p), ("in_ln_w", ctypes.c_void_p),
                ("layer", ctypes.POINTER(DgsDitLayerWeights)),
                ("ada_w", ctypes.c_void_p), ("ada_b", ctypes.c_void_p),
                ("up_ln_w", ctypes.c_void_p), ("up_w", ctypes.c_void_p), ("dec_ln_w", ctypes.c_void_p), ("dec_w", ctypes.c_void_p)]


class DgsDitForwardArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("V", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("images", ctypes.c_void_p), ("ray_o", ctypes.c_void_p), ("ray_d", ctypes.c_void_p), ("t", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("xyz", ctypes.c_void_p), ("features", ctypes.c_void_p), ("scaling", ctypes.c_void_p),
                ("rotation", ctypes.c_void_p), ("opacity", ctypes.c_void_p), ("aligned_xyz", ctypes.c_void_p),
                ("tokens", ctypes.c_void_p),
                ("prof_events", ctypes.POINTER(ctypes.c_void_p)), ("prof_kind", ctypes.c_int32),
                ("prof_capacity", ctypes.c_int32), ("prof_count", ctypes.POINTER(ctypes.c_int32)),
                ("train_recompute", ctypes.c_int32)]


class DgsDitLayerWeightsT(ctypes.Structure):
    _fields_ = [("qkv_wT", ctypes.c_void_p), ("proj_wT", ctypes.c_void_p), ("fc1_wT", ctypes.c_void_p), ("fc2_wT", ctypes.c_void_p)]


class DgsDitModelT(ctypes.Structure):
    _fields_ = [("layer", ctypes.POINTER(DgsDitLayerWeightsT)), ("dec_wT", ctypes.c_void_p)]


class DgsDitLayerGrads(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("qkv_w", "proj_w", "fc1_w", "fc2_w", "qkv_b", "proj_b", "fc1_b", "fc2_b", "ada_w", "ada_b")]


class DgsDitGrads(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("t_w0", "t_b0", "t_w1", "t_b1", "tok_w", "pos_emb", "in_ln_w")] + \
               [("layer", ctypes.POINTER(DgsDitLayerGrads))] + \
               [(k, ctypes.c_void_p) for k in ("head_ada_w", "head_ada_b", "up_ln_w", "up_w", "dec_ln_w", "dec_w")]


class DgsDitBackwardArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("V", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("ray_d", ctypes.c_void_p), ("saved", ctypes.c_void_p), ("saved_bytes", ctypes.c_size_t),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("dxyz", ctypes.c_void_p), ("dfeatures", ctypes.c_void_p), ("dscaling", ctypes.c_void_p),
                ("drotation", ctypes.c_void_p), ("dopacity", ctypes.c_void_p), ("recompute", ctypes.c_int32),
                ("block_done", ctypes.c_void_p), ("block_user", ctypes.c_void_p)]


BLOCK_DONE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int32)


class DgsDitRunBlocksArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("L", ctypes.c_int32), ("V", ctypes.c_int32), ("first", ctypes.c_int32), ("last", ctypes.c_int32),
                ("tokens_in", ctypes.c_void_p), ("cvec", ctypes.c_void_p), ("tokens_out", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t)]


# every symbol include/dgs_dit.h declares (checked by tests/test_abi.py)
DIT_SYMBOLS = ["dgs_dit_gemm", "dgs_dit_attention", "dgs_dit_attention_backward", "dgs_dit_layernorm_backward",
               "dgs_dit_rowlinear_backward", "dgs_dit_gate_mul", "dgs_dit_saved_bytes", "dgs_dit_backward_workspace_bytes",
               "dgs_dit_forward_train", "dgs_dit_backward", "dgs_dit_layernorm", "dgs_dit_rowlinear", "dgs_dit_lpad",
               "dgs_dit_workspace_bytes", "dgs_dit_forward", "dgs_dit_gemm_splitk_bytes",
               "dgs_dit_attention_tail_bytes", "dgs_dit_run_blocks", "dgs_debug_poison_lds", "dgs_debug_clock_probe",
               "dgs_dit_layernorm_backward_scratch_bytes", "dgs_dit_rowlinear_backward_scratch_bytes", "dgs_dit_gate_mul_scratch_bytes",
               "dgs_dit_workspace_bytes_for_tokens", "dgs_dit_attention_backward_slots", "dgs_dit_layernorm_gemm", "dgs_dit_layernorm_gemm_shares_rows", "dgs_dit_gemm_sliced_tile"]


def _declare_dit(L):
    L.dgs_dit_gemm_sliced_tile.restype = ctypes.c_int32
    L.dgs_dit_gemm_sliced_tile.argtypes = [ctypes.POINTER(DgsDitGemmArgs)]
    L.dgs_dit_layernorm_gemm_shares_rows.restype = ctypes.c_int32
    L.dgs_dit_layernorm_gemm_shares_rows.argtypes = [ctypes.POINTER(DgsDitLayerNormArgs), ctypes.POINTER(DgsDitGemmArgs)]
    L.dgs_dit_layernorm_gemm.restype = ctypes.c_int
    L.dgs_dit_layernorm_gemm.argtypes = [ctypes.POINTER(DgsDitLayerNormArgs), ctypes.POINTER(DgsDitGemmArgs), ctypes.c_void_p]
    L.dgs_dit_attention_backward_slots.restype = ctypes.c_int32
    L.dgs_dit_attention_backward_slots.argtypes = [ctypes.c_int32]
    for name, argt in (("dgs_dit_gemm", DgsDitGemmArgs), ("dgs_dit_attention", DgsDitAttentionArgs),
                       ("dgs_dit_attention_backward", DgsDitAttentionBackwardArgs),
                       ("dgs_dit_layernorm_backward", DgsDitLayerNormBackwardArgs),
                       ("dgs_dit_rowlinear_backward", DgsDitRowLinearBackwardArgs), ("dgs_dit_gate_mul", DgsDitGateMulArgs),
                       ("dgs_dit_layernorm", DgsDitLayerNormArgs), ("dgs_dit_rowlinear", DgsDitRowLinearArgs)):
        fn = getattr(L, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(argt), ctypes.c_void_p]
    for fn in (L.dgs_dit_layernorm_backward_scratch_bytes, L.dgs_dit_rowlinear_backward_scratch_bytes, L.dgs_dit_gate_mul_scratch_bytes):
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.c_int32] * 3
    L.dgs_dit_gemm_splitk_bytes.restype = ctypes.c_size_t
    L.dgs_dit_gemm_splitk_bytes.argtypes = [ctypes.c_int32] * 4
    L.dgs_dit_attention_tail_bytes.restype = ctypes.c_size_t
    L.dgs_dit_attention_tail_bytes.argtypes = [ctypes.c_int32] * 3
    L.dgs_dit_lpad.restype = ctypes.c_int32
    L.dgs_dit_lpad.argtypes = [ctypes.c_int32]
    L.dgs_dit_workspace_bytes.restype = ctypes.c_size_t
    L.dgs_dit_workspace_bytes.argtypes = [ctypes.POINTER(DgsDitModel), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    L.dgs_dit_workspace_bytes_for_tokens.restype = ctypes.c_size_t
    L.dgs_dit_workspace_bytes_for_tokens.argtypes = [ctypes.POINTER(DgsDitModel), ctypes.c_int32, ctypes.c_int32]
    for fn in (L.dgs_dit_saved_bytes, L.dgs_dit_backward_workspace_bytes):
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.POINTER(DgsDitModel), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    L.dgs_dit_saved_bytes.argtypes = L.dgs_dit_saved_bytes.argtypes + [ctypes.c_int32]
    L.dgs_debug_poison_lds.restype = ctypes.c_int
    L.dgs_debug_poison_lds.argtypes = [ctypes.c_void_p]
    L.dgs_debug_clock_probe.restype = ctypes.c_int
    L.dgs_debug_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    L.dgs_dit_run_blocks.restype = ctypes.c_int
    L.dgs_dit_run_blocks.argtypes = [ctypes.POINTER(DgsDitModel), ctypes.POINTER(DgsDitRunBlocksArgs), ctypes.c_void_p]
    L.dgs_dit_forward_train.restype = ctypes.c_int
    L.dgs_dit_forward_train.argtypes = [ctypes.POINTER(DgsDitModel), ctypes.POINTER(DgsDitForwardArgs), ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.c_void_p]
    L.dgs_dit_backward.restype = ctypes.c_int
    L.dgs_dit_backward.argtypes = [ctypes.POINTER(DgsDitModel), ctypes.POINTER(DgsDitModelT), ctypes.POINTER(DgsDitGrads),
                                   ctypes.POINTER(DgsDitBackwardArgs), ctypes.c_void_p]
    L.dgs_dit_forward.restype = ctypes.c_int
    L.dgs_dit_forward.argtypes = [ctypes.POINTER(DgsDitModel), ctypes.POINTER(DgsDitForwardArgs), ctypes.c_void_p]
    return L


class DgsSamplerStepArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("per_sample", ctypes.c_int64), ("model_stride", ctypes.c_int64), ("model_offset", ctypes.c_int64),
                ("model_output", ctypes.c_void_p), ("x_t", ctypes.c_void_p), ("noise", ctypes.c_void_p), ("t", ctypes.c_void_p),
                ("coef1", ctypes.c_void_p), ("coef2", ctypes.c_void_p), ("sigma", ctypes.c_void_p), ("T", ctypes.c_int32),
                ("clip_denoised", ctypes.c_int32), ("out", ctypes.c_void_p), ("pred_xstart", ctypes.c_void_p), ("bad_t", ctypes.c_void_p)]


# every symbol include/dgs_sampler.h declares (checked by tests/test_abi.py)
SAMPLER_SYMBOLS = ["dgs_sampler_step"]


def _declare_sampler(L):
    L.dgs_sampler_step.restype = ctypes.c_int
    L.dgs_sampler_step.argtypes = [ctypes.POINTER(DgsSamplerStepArgs), ctypes.c_void_p]
    return L


class DgsMseArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("n", ctypes.c_int64), ("rendering", ctypes.c_void_p), ("target", ctypes.c_void_p),
                ("clamp01", ctypes.c_int32), ("l2", ctypes.c_void_p), ("psnr", ctypes.c_void_p), ("grad", ctypes.c_void_p),
                ("grad_scale", ctypes.c_float), ("partial", ctypes.c_void_p)]


LOSS_CHUNKS = 64
# every symbol include/dgs_loss.h declares (checked by tests/test_abi.py)
class DgsResizeArgs(ctypes.Structure):
    _fields_ = [("planes", ctypes.c_int32), ("in_h", ctypes.c_int32), ("in_w", ctypes.c_int32), ("out_h", ctypes.c_int32),
                ("out_w", ctypes.c_int32), ("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("mul", ctypes.c_float),
                ("add", ctypes.c_float), ("ddst", ctypes.c_void_p), ("dsrc", ctypes.c_void_p)]


class DgsPointsLossArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("V", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("aligned", ctypes.c_void_p),
                ("ray_o", ctypes.c_void_p), ("gt", ctypes.c_void_p), ("masks", ctypes.c_void_p), ("pointsdist", ctypes.c_void_p),
                ("xyz", ctypes.c_void_p), ("w_pointsdist", ctypes.c_void_p), ("w_xyz", ctypes.c_float), ("grad", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p)]


LOSS_SYMBOLS = ["dgs_mse_psnr", "dgs_resize_bilinear", "dgs_resize_bilinear_backward", "dgs_points_loss", "dgs_points_loss_workspace_floats"]


def _declare_loss(L):
    L.dgs_mse_psnr.restype = ctypes.c_int
    L.dgs_mse_psnr.argtypes = [ctypes.POINTER(DgsMseArgs), ctypes.c_void_p]
    for fn in (L.dgs_resize_bilinear, L.dgs_resize_bilinear_backward):
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(DgsResizeArgs), ctypes.c_void_p]
    L.dgs_points_loss.restype = ctypes.c_int
    L.dgs_points_loss.argtypes = [ctypes.POINTER(DgsPointsLossArgs), ctypes.c_void_p]
    L.dgs_points_loss_workspace_floats.restype = ctypes.c_int64
    L.dgs_points_loss_workspace_floats.argtypes = [ctypes.c_int32, ctypes.c_int32]
    return L


class DgsAdamWTensor(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("copy", ctypes.c_void_p),
                ("copy_t", ctypes.c_void_p), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("copy_kind", ctypes.c_int32),
                ("first_tile", ctypes.c_int32)]


class DgsAdamWArgs(ctypes.Structure):
    _fields_ = [("tensors", ctypes.c_void_p), ("n_tensors", ctypes.c_int32), ("n_tiles", ctypes.c_int32), ("lr", ctypes.c_float),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float),
                ("bias_correction1", ctypes.c_float), ("bias_correction2_sqrt", ctypes.c_float),
                ("grad_sumsq", ctypes.c_void_p), ("max_grad_norm", ctypes.c_float)]


OPTIM_COPY_NONE, OPTIM_COPY_BF16, OPTIM_COPY_F32 = 0, 1, 2
# every symbol include/dgs_optim.h declares (checked by tests/test_abi.py)
OPTIM_SYMBOLS = ["dgs_adamw_plan", "dgs_adamw_step", "dgs_sumsq_count", "dgs_sumsq_partials", "dgs_sumsq_finish"]


def _declare_optim(L):
    L.dgs_adamw_plan.restype = ctypes.c_int32
    L.dgs_adamw_plan.argtypes = [ctypes.POINTER(DgsAdamWTensor), ctypes.c_int32]
    L.dgs_adamw_step.restype = ctypes.c_int
    L.dgs_adamw_step.argtypes = [ctypes.POINTER(DgsAdamWArgs), ctypes.c_void_p]
    L.dgs_sumsq_count.restype = ctypes.c_int32
    L.dgs_sumsq_count.argtypes = [ctypes.c_int64]
    L.dgs_sumsq_partials.restype = ctypes.c_int
    L.dgs_sumsq_partials.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    L.dgs_sumsq_finish.restype = ctypes.c_int
    L.dgs_sumsq_finish.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    return L


_declare_raster = _declare


def _declare(L):  # noqa: F811  (raster + DiT + sampler + loss prototypes on one library)
    return _declare_optim(_declare_loss(_declare_sampler(_declare_dit(_declare_raster(L)))))
