"""ctypes binding of the C ABI declared in include/b3gs_raster.h.

There is no fallback: if libb3gs_raster.so is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("B3GS_LIB") or os.path.join(_HERE, "libb3gs_raster.so")   # B3GS_LIB: A/B builds of the kernels

ABI_VERSION = 10
OK = 0
ERR_NAMES = {-1: "B3GS_ERR_ARG", -2: "B3GS_ERR_ALLOC", -3: "B3GS_ERR_HIP", -4: "B3GS_ERR_CAPACITY",
             -5: "B3GS_ERR_NO_DEVICE"}

c_float_p = C.c_void_p  # device pointers travel as plain integers

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class B3gsScene(C.Structure):
    _fields_ = [("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("scale_modifier", C.c_float),
                ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p)]


class B3gsRawParams(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("features_dc", C.c_void_p), ("features_rest", C.c_void_p),
                ("scaling", C.c_void_p), ("rotation", C.c_void_p), ("opacity", C.c_void_p)]


class B3gsRawGrads(C.Structure):
    _fields_ = B3gsRawParams._fields_ + [("touched_rows", C.c_void_p)]


class B3gsFusedView(C.Structure):
    _fields_ = [("view", C.POINTER(B3gsScene)), ("radii", C.c_void_p), ("geometry", C.c_void_p),
                ("scratch", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("densify_stats", C.c_int32)]


class B3gsBlendView(C.Structure):
    _fields_ = [("view", C.POINTER(B3gsScene)), ("geometry", C.c_void_p), ("binning", C.c_void_p),
                ("image", C.c_void_p), ("out_color", C.c_void_p), ("out_depth", C.c_void_p),
                ("out_alpha", C.c_void_p), ("dL_dcolor", C.c_void_p), ("dL_ddepth", C.c_void_p),
                ("dL_dalpha", C.c_void_p), ("scratch", C.c_void_p), ("binning_capacity", C.c_int64)]


class B3gsForwardView(C.Structure):
    _fields_ = [("view", C.POINTER(B3gsScene)), ("geometry", C.c_void_p), ("binning", C.c_void_p),
                ("binning_capacity", C.c_int64), ("image", C.c_void_p), ("out_color", C.c_void_p),
                ("out_depth", C.c_void_p), ("out_alpha", C.c_void_p), ("radii", C.c_void_p),
                ("device_num_rendered", C.c_void_p), ("depth_order_from", C.c_int32), ("seg1_fraction", C.c_float),
                ("high_water", C.c_void_p), ("overflow_flag", C.c_void_p), ("depth_key_bits", C.c_int32),
                ("fresh_image", C.c_int32), ("depth_order_hint", C.c_void_p), ("hint_mismatch", C.c_void_p),
                ("hint_trusted", C.c_int32), ("visible", C.c_void_p), ("reference_binning", C.c_int32)]


class B3gsLossIO(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("image", C.c_void_p), ("depth", C.c_void_p), ("alpha", C.c_void_p),
                ("gt_image", C.c_void_p), ("shifted_image", C.c_void_p), ("alpha_weight", C.c_void_p),
                ("focal_x", C.c_float), ("trans_dist", C.c_float), ("lambda_dssim", C.c_float),
                ("lambda_smooth", C.c_float), ("grad_scale", C.c_float), ("dL_dimage", C.c_void_p),
                ("dL_ddepth", C.c_void_p), ("dL_dalpha", C.c_void_p), ("dL_dshifted", C.c_void_p),
                ("parts", C.c_void_p), ("workspace", C.c_void_p), ("trans_dist_dev", C.c_void_p)]


class B3gsDensifyIO(C.Structure):
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("param", C.c_void_p * 6), ("exp_avg", C.c_void_p * 6),
                ("exp_avg_sq", C.c_void_p * 6), ("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p),
                ("grad_threshold", C.c_float), ("min_opacity", C.c_float), ("extent", C.c_float),
                ("percent_dense", C.c_float), ("max_screen_size", C.c_float)]


class B3gsAdamSegment(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("count", C.c_int64), ("lr", C.c_float), ("row_len", C.c_int32), ("first_row", C.c_int32),
                ("lr_dev", C.c_void_p)]


class B3gsDensifyStats(C.Structure):
    _fields_ = [("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p), ("max_radii2D", C.c_void_p),
                ("skip_if_nonzero", C.c_void_p)]


class B3gsDebugViews(C.Structure):
    _fields_ = [("tiles_touched", C.c_void_p), ("depths", C.c_void_p), ("records", C.c_void_p),
                ("point_list", C.c_void_p), ("tile_ids", C.c_void_p), ("ranges", C.c_void_p),
                ("final_T", C.c_void_p), ("n_contrib", C.c_void_p), ("packed_idx_bits", C.c_int32),
                ("counts", C.c_void_p), ("point_list2", C.c_void_p), ("ranges2", C.c_void_p)]


class B3gsKernelTimes(C.Structure):
    _fields_ = [("preprocess_ms", C.c_double), ("sort_ms", C.c_double), ("render_fwd_ms", C.c_double),
                ("render_bwd_ms", C.c_double), ("preprocess_bwd_ms", C.c_double), ("calls", C.c_int64)]


# every symbol include/b3gs_raster.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTS = ("b3gs_abi_version", "b3gs_last_error", "b3gs_set_timing", "b3gs_timing_collect", "b3gs_geometry_bytes", "b3gs_image_bytes",
           "b3gs_binning_bytes", "b3gs_forward", "b3gs_forward_capacity", "b3gs_backward", "b3gs_mark_visible",
           "b3gs_debug_views", "b3gs_forward_raw", "b3gs_backward_raw", "b3gs_backward_scratch_floats",
           "b3gs_backward_raw_accumulate", "b3gs_blend_forward_batch", "b3gs_blend_backward_batch",
           "b3gs_adam_step", "b3gs_forward_raw_batch", "b3gs_loss_workspace_floats", "b3gs_binocular_loss",
           "b3gs_densify_classify", "b3gs_densify_scatter", "b3gs_knn_workspace_bytes", "b3gs_knn_mean_dist2",
           "b3gs_backward_raw_accumulate_range", "b3gs_binocular_loss_batch",
           # ABI 9: the training loop's statements one by one
           "b3gs_lossfn_workspace_floats", "b3gs_l1_loss_forward", "b3gs_l1_loss_backward", "b3gs_inverse_warp_forward",
           "b3gs_inverse_warp_backward", "b3gs_smooth_loss_forward", "b3gs_smooth_loss_backward", "b3gs_ssim_forward",
           "b3gs_ssim_backward", "b3gs_opacity_decay", "b3gs_add_densification_stats", "b3gs_adam_step_at", "b3gs_debug_activations",
           "b3gs_apply_staged_densify_stats")

_lib = None


class B3gsError(RuntimeError):
    pass


def lib():
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise B3gsError(f"{LIB_PATH} not found: run `python -m binocular3dgs_amd.build` (hipcc, gfx950). "
                        "There is no CPU or PyTorch fallback for the rasterizer.")
    L = C.CDLL(LIB_PATH)
    L.b3gs_abi_version.restype = C.c_int
    L.b3gs_last_error.restype = C.c_char_p
    L.b3gs_set_timing.argtypes = [C.c_void_p]
    L.b3gs_set_timing.restype = None
    L.b3gs_timing_collect.restype = C.c_int
    L.b3gs_geometry_bytes.argtypes = [C.c_int32]
    L.b3gs_geometry_bytes.restype = C.c_size_t
    L.b3gs_image_bytes.argtypes = [C.c_int32, C.c_int32]
    L.b3gs_image_bytes.restype = C.c_size_t
    L.b3gs_binning_bytes.argtypes = [C.c_int32, C.c_int64]
    L.b3gs_binning_bytes.restype = C.c_size_t
    L.b3gs_forward.argtypes = [C.POINTER(B3gsScene), ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int32), C.c_void_p]
    L.b3gs_forward.restype = C.c_int
    L.b3gs_forward_capacity.argtypes = [C.POINTER(B3gsScene), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.b3gs_forward_capacity.restype = C.c_int
    L.b3gs_backward.argtypes = [C.POINTER(B3gsScene), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_void_p] * 8 + [C.c_void_p]
    L.b3gs_backward.restype = C.c_int
    L.b3gs_forward_raw.argtypes = [C.POINTER(B3gsScene), C.POINTER(B3gsRawParams), C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p]
    L.b3gs_forward_raw.restype = C.c_int
    for fn in (L.b3gs_blend_forward_batch, L.b3gs_blend_backward_batch):
        fn.argtypes = [C.c_int32, C.POINTER(B3gsBlendView), C.c_void_p]
        fn.restype = C.c_int
    L.b3gs_backward_scratch_floats.argtypes = [C.c_int32]
    L.b3gs_backward_scratch_floats.restype = C.c_size_t
    L.b3gs_backward_raw.argtypes = [C.POINTER(B3gsScene), C.POINTER(B3gsRawParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(B3gsRawGrads), C.c_void_p, C.c_int, C.c_void_p]
    L.b3gs_backward_raw.restype = C.c_int
    L.b3gs_backward_raw_accumulate.argtypes = [C.c_int32, C.POINTER(B3gsFusedView), C.POINTER(B3gsRawParams),
                                               C.POINTER(B3gsRawGrads), C.c_int32, C.POINTER(B3gsDensifyStats),
                                               C.c_void_p]
    L.b3gs_backward_raw_accumulate.restype = C.c_int
    L.b3gs_backward_raw_accumulate_range.argtypes = [C.c_int32, C.POINTER(B3gsFusedView), C.POINTER(B3gsRawParams),
                                                     C.POINTER(B3gsRawGrads), C.c_int32, C.POINTER(B3gsDensifyStats),
                                                     C.c_int32, C.c_int32, C.c_void_p]
    L.b3gs_backward_raw_accumulate_range.restype = C.c_int
    L.b3gs_forward_raw_batch.argtypes = [C.c_int32, C.POINTER(B3gsForwardView), C.POINTER(B3gsRawParams), C.c_int,
                                         C.c_void_p]
    L.b3gs_forward_raw_batch.restype = C.c_int
    L.b3gs_loss_workspace_floats.argtypes = [C.c_int32, C.c_int32]
    L.b3gs_loss_workspace_floats.restype = C.c_size_t
    L.b3gs_binocular_loss.argtypes = [C.POINTER(B3gsLossIO), C.c_void_p]
    L.b3gs_binocular_loss.restype = C.c_int
    L.b3gs_binocular_loss_batch.argtypes = [C.c_int32, C.POINTER(B3gsLossIO), C.c_void_p]
    L.b3gs_binocular_loss_batch.restype = C.c_int
    L.b3gs_densify_classify.argtypes = [C.POINTER(B3gsDensifyIO), C.c_void_p, C.c_void_p]
    L.b3gs_densify_classify.restype = C.c_int
    L.b3gs_densify_scatter.argtypes = [C.POINTER(B3gsDensifyIO), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.c_void_p]
    L.b3gs_densify_scatter.restype = C.c_int
    L.b3gs_knn_workspace_bytes.argtypes = [C.c_int32]
    L.b3gs_knn_workspace_bytes.restype = C.c_size_t
    L.b3gs_knn_mean_dist2.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.b3gs_knn_mean_dist2.restype = C.c_int
    L.b3gs_adam_step.argtypes = [C.c_int32, C.POINTER(B3gsAdamSegment), C.c_void_p, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.b3gs_adam_step.restype = C.c_int
    V, I32, I64, F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.b3gs_lossfn_workspace_floats.argtypes = [I64, I32, I32]
    L.b3gs_lossfn_workspace_floats.restype = C.c_size_t
    L.b3gs_l1_loss_forward.argtypes = [V, V, V, I64, I32, I64, V, V, V]
    L.b3gs_l1_loss_backward.argtypes = [V, V, V, I64, I32, I64, V, V, V, V, V]
    L.b3gs_inverse_warp_forward.argtypes = [V, V, I32, I32, I32, I32, V, V, V]
    L.b3gs_inverse_warp_backward.argtypes = [V, V, V, I32, I32, I32, I32, V, V, V]
    L.b3gs_smooth_loss_forward.argtypes = [V, V, I32, I32, I32, I32, V, V, V]
    L.b3gs_smooth_loss_backward.argtypes = [V, V, I32, I32, I32, I32, V, V, V, V]
    L.b3gs_ssim_forward.argtypes = [V, V, I32, I32, I32, I32, I32, V, I32, V, V, V]
    L.b3gs_ssim_backward.argtypes = [V, V, V, I32, I32, I32, I32, I32, V, V, V, V]
    L.b3gs_opacity_decay.argtypes = [V, I64, F, V]
    L.b3gs_add_densification_stats.argtypes = [I64, V, I64, V, V, V, V]
    L.b3gs_adam_step_at.argtypes = [I32, C.POINTER(B3gsAdamSegment), I32, F, F, F, V]
    for fn in (L.b3gs_l1_loss_forward, L.b3gs_l1_loss_backward, L.b3gs_inverse_warp_forward, L.b3gs_inverse_warp_backward,
               L.b3gs_smooth_loss_forward, L.b3gs_smooth_loss_backward, L.b3gs_ssim_forward, L.b3gs_ssim_backward,
               L.b3gs_opacity_decay, L.b3gs_add_densification_stats, L.b3gs_adam_step_at):
        fn.restype = C.c_int
    L.b3gs_apply_staged_densify_stats.argtypes = [I64] + [V] * 9
    L.b3gs_apply_staged_densify_stats.restype = C.c_int
    L.b3gs_debug_activations.argtypes = [I32, C.POINTER(B3gsRawParams), V, V, V, V]
    L.b3gs_debug_activations.restype = C.c_int
    L.b3gs_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.b3gs_mark_visible.restype = C.c_int
    L.b3gs_debug_views.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(B3gsDebugViews)]
    L.b3gs_debug_views.restype = C.c_int
    if L.b3gs_abi_version() != ABI_VERSION:
        raise B3gsError(f"libb3gs_raster.so ABI {L.b3gs_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != OK:
        msg = lib().b3gs_last_error().decode("utf-8", "replace")
        raise B3gsError(f"{what} failed: {ERR_NAMES.get(rc, rc)}: {msg}")
