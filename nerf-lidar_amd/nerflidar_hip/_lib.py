"""ctypes binding of libnerflidar_hip.so (the C ABI of include/nerflidar_hip.h).

There is no CPU fallback: if the shared object is missing or a call fails, this raises.  Build it
with `make -C nerf-lidar_amd` (or `python -c "import __graft_entry__ as g; g.build()"`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NLR_LIB_PATH") or os.path.join(_HERE, "libnerflidar_hip.so")  # override: diagnostic builds only

NLR_MAX_LEVELS = 4
NLR_MAX_VIEW_DEPTH = 16
PREC_F32, PREC_MIXED, PREC_FAST = 0, 1, 2

c_fp = C.c_void_p  # device / host pointers travel as integers


class NlrLinear(C.Structure):
    _fields_ = [("weight", c_fp), ("bias", c_fp), ("out_features", C.c_uint32), ("in_features", C.c_uint32)]


class NlrGridDesc(C.Structure):
    _fields_ = [("table", c_fp), ("table_dtype", C.c_int32), ("num_levels", C.c_uint32), ("level_dim", C.c_uint32),
                ("base_resolution", C.c_uint32), ("log2_per_level_scale", C.c_float), ("offsets", c_fp),
                ("gridtype", C.c_uint32), ("align_corners", C.c_uint32), ("interp", C.c_uint32)]


class NlrMlpDesc(C.Structure):
    _fields_ = [("grid", NlrGridDesc), ("density0", NlrLinear), ("density2", NlrLinear), ("disable_rgb", C.c_uint32),
                ("bottleneck_width", C.c_uint32), ("net_depth_viewdirs", C.c_uint32), ("net_width_viewdirs", C.c_uint32),
                ("skip_layer_dir", C.c_uint32), ("deg_view", C.c_uint32), ("view", NlrLinear * NLR_MAX_VIEW_DEPTH),
                ("rgb_layer", NlrLinear), ("use_semantic", C.c_uint32), ("no_sem_layer", C.c_uint32),
                ("class_num", C.c_uint32), ("sem0", NlrLinear), ("sem2", NlrLinear), ("use_intensity", C.c_uint32),
                ("int0", NlrLinear), ("int2", NlrLinear), ("density_bias", C.c_float), ("rgb_premultiplier", C.c_float),
                ("rgb_bias", C.c_float), ("rgb_padding", C.c_float), ("re_weights", C.c_uint32)]


class NlrObjClassDesc(C.Structure):
    _fields_ = [("mlp", NlrMlpDesc), ("latent_size", C.c_uint32), ("split_latent", C.c_uint32), ("class_type", C.c_int32)]


class NlrObjectsDesc(C.Structure):
    _fields_ = [("n_classes", C.c_uint32), ("classes", C.POINTER(NlrObjClassDesc)), ("n_tracks", C.c_uint32), ("track_class", c_fp),
                ("latents", c_fp)]


class NlrModelDesc(C.Structure):
    _fields_ = [("num_levels", C.c_uint32), ("num_samples", C.c_uint32 * NLR_MAX_LEVELS),
                ("mlps", C.POINTER(NlrMlpDesc) * NLR_MAX_LEVELS), ("dilation_multiplier", C.c_float),
                ("dilation_bias", C.c_float), ("anneal_slope", C.c_float), ("resample_padding", C.c_float),
                ("power_lambda", C.c_float), ("std_scale", C.c_float), ("bg_intensity", C.c_float),
                ("opaque_background", C.c_uint32), ("mlp_precision", C.c_uint32)]


class NlrRays(C.Structure):
    _fields_ = [(k, c_fp) for k in ("origins", "directions", "viewdirs", "radii", "near", "far", "base_x", "base_y")]


class NlrRenderCfg(C.Structure):
    _fields_ = [("train_frac", C.c_float), ("compute_extras", C.c_uint32), ("sample_n", C.c_uint32),
                ("sample_m", C.c_uint32), ("rand_jitter", c_fp * NLR_MAX_LEVELS), ("rand_deg", c_fp * NLR_MAX_LEVELS),
                ("scale_factor", C.c_float), ("shuffled_rays", C.c_uint32)]


class NlrLevelOut(C.Structure):
    _fields_ = [(k, c_fp) for k in ("sdist", "tdist", "weights", "density", "rgb", "semantic", "intensity", "depth", "r_rgb", "r_acc",
                                    "r_distance_mean", "r_distance_median", "r_distance_percentile_5", "r_distance_percentile_95")]


class NlrOut(C.Structure):
    _fields_ = [(k, c_fp) for k in ("rgb", "depth", "semantic", "intensity", "acc", "distance_mean", "distance_median",
                                    "distance_percentile_5", "distance_percentile_95", "labels", "points", "packed")] + \
               [("packed_h", C.c_uint32), ("packed_w", C.c_uint32), ("history", NlrLevelOut * NLR_MAX_LEVELS)]


_lib = None

EXPORTS = ["nlr_last_error", "nlr_version", "nlr_build_sha", "nlr_debug_set", "nlr_debug_get", "nlr_grid_fast_path", "nlr_grid_encode_forward", "nlr_grid_encode_backward", "nlr_grad_total_variation", "nlr_level_scale",
           "nlr_sample_u", "nlr_model_create", "nlr_model_destroy", "nlr_model_set_table", "nlr_workspace_bytes",
           "nlr_render_rays", "nlr_kernel_names", "nlr_resample_level", "nlr_mlp_level", "nlr_composite_level",
           "nlr_profile_begin", "nlr_profile_begin_kinds", "nlr_profile_end", "nlr_range_workspace_bytes", "nlr_range_project",
           "nlr_composite_backward", "nlr_hash_decay_forward", "nlr_hash_decay_backward", "nlr_box_winner", "nlr_cast_contract",
           "nlr_track_box_params", "nlr_objects_create", "nlr_objects_destroy", "nlr_objects_workspace_bytes", "nlr_objects_apply",
           "nlr_render_rays_dynamic", "nlr_prop_mlp_forward", "nlr_prop_mlp_backward", "nlr_encode_features_forward",
           "nlr_encode_features_backward", "nlr_encode_features_backward_ws", "nlr_grid_encode_backward_ws", "nlr_grid_backward_workspace_bytes",
           "nlr_train_plan_create", "nlr_train_plan_destroy", "nlr_train_act_width", "nlr_train_param_layout", "nlr_train_pack",
           "nlr_mlp_train_forward", "nlr_mlp_train_backward"]
NLR_K_COUNT = 6
DBG_FORCE_GENERIC, DBG_MLP_WORKGROUPS, DBG_BINNED_C4, DBG_NO_XPAIR_SCATTER, DBG_SCATTER_LEVELS, DBG_NO_SCATTER_CACHE, DBG_RAY_GROUPS = 0, 1, 2, 3, 4, 5, 6


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension has not been built (make -C nerf-lidar_amd). "
                               "There is no CPU fallback for the render path.")
        L = C.CDLL(LIB_PATH)
        L.nlr_last_error.restype = C.c_char_p
        L.nlr_kernel_names.restype = C.c_char_p
        L.nlr_build_sha.restype = C.c_char_p
        L.nlr_version.restype = C.c_int
        L.nlr_workspace_bytes.restype = C.c_size_t
        L.nlr_workspace_bytes.argtypes = [c_fp, C.c_uint32]
        L.nlr_model_destroy.restype = None
        L.nlr_model_destroy.argtypes = [c_fp]
        L.nlr_model_create.argtypes = [C.POINTER(NlrModelDesc), C.POINTER(c_fp), c_fp]
        L.nlr_model_set_table.argtypes = [c_fp, C.c_uint32, c_fp, C.c_int]
        L.nlr_level_scale.restype = None
        L.nlr_level_scale.argtypes = [C.c_uint32, C.c_float, C.c_uint32, c_fp, c_fp]
        L.nlr_sample_u.restype = None
        L.nlr_sample_u.argtypes = [C.c_uint32, C.c_int, c_fp, c_fp]
        L.nlr_grid_encode_forward.argtypes = [c_fp, c_fp, C.c_int, c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_float, C.c_uint32, c_fp, C.c_uint32, C.c_int, C.c_uint32,
                                              C.c_int, c_fp]
        L.nlr_grid_encode_backward.argtypes = [c_fp, c_fp, c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_float, C.c_uint32, c_fp, c_fp, C.c_uint32, C.c_int, C.c_uint32,
                                               C.c_int, c_fp]
        L.nlr_grid_encode_backward_ws.argtypes = L.nlr_grid_encode_backward.argtypes[:-1] + [c_fp, C.c_size_t, c_fp]
        L.nlr_grid_backward_workspace_bytes.restype = C.c_size_t
        L.nlr_grid_backward_workspace_bytes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, c_fp, C.c_uint32, C.c_int]
        L.nlr_debug_set.argtypes = [C.c_uint32, C.c_int]
        L.nlr_debug_get.argtypes = [C.c_uint32]
        L.nlr_grid_fast_path.argtypes = [c_fp, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32]
        L.nlr_grad_total_variation.argtypes = [c_fp, c_fp, c_fp, c_fp, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                               C.c_uint32, C.c_uint32, C.c_int, c_fp]
        L.nlr_render_rays.argtypes = [c_fp, C.POINTER(NlrRays), C.c_uint32, C.POINTER(NlrRenderCfg), C.POINTER(NlrOut),
                                      c_fp, C.c_size_t, c_fp]
        L.nlr_resample_level.argtypes = [c_fp, c_fp, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, c_fp, c_fp,
                                         c_fp, C.c_float, C.c_uint32, c_fp, c_fp, c_fp]
        L.nlr_mlp_level.argtypes = [c_fp, C.c_uint32, C.POINTER(NlrRays), c_fp, C.c_uint32, C.c_uint32, C.c_uint32, c_fp,
                                    c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_size_t, c_fp]
        L.nlr_composite_level.argtypes = [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_int, C.c_float, C.c_int, C.c_float, c_fp, C.POINTER(NlrOut),
                                          c_fp, c_fp]
        L.nlr_range_workspace_bytes.restype = C.c_size_t
        L.nlr_range_workspace_bytes.argtypes = [C.c_uint32, C.c_uint32]
        L.nlr_range_project.argtypes = [c_fp, c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, c_fp,
                                        C.c_size_t, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
        L.nlr_composite_backward.argtypes = [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                             C.c_float, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
        L.nlr_hash_decay_forward.argtypes = [c_fp, c_fp, C.c_uint32, C.c_uint32, c_fp, c_fp]
        L.nlr_hash_decay_backward.argtypes = [c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_float, c_fp, c_fp]
        L.nlr_cast_contract.argtypes = [C.POINTER(NlrRays), c_fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, c_fp, c_fp, c_fp,
                                        c_fp]
        L.nlr_box_winner.argtypes = [c_fp, c_fp, c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_uint32, c_fp, c_fp]
        L.nlr_track_box_params.argtypes = [c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_uint32, c_fp, c_fp]
        L.nlr_objects_create.argtypes = [C.POINTER(NlrObjectsDesc), C.POINTER(c_fp), c_fp]
        L.nlr_objects_destroy.restype = None
        L.nlr_objects_destroy.argtypes = [c_fp]
        L.nlr_objects_workspace_bytes.restype = C.c_size_t
        L.nlr_objects_workspace_bytes.argtypes = [c_fp, C.c_uint32, C.c_uint32]
        L.nlr_objects_apply.argtypes = [c_fp, C.POINTER(NlrRays), c_fp, c_fp, C.c_uint32, C.c_uint32, C.c_uint32, c_fp, c_fp, c_fp,
                                        C.c_uint32, c_fp, c_fp, C.c_size_t, c_fp]
        L.nlr_render_rays_dynamic.argtypes = [c_fp, c_fp, C.POINTER(NlrRays), c_fp, C.c_uint32, C.c_uint32, C.POINTER(NlrRenderCfg),
                                              C.POINTER(NlrOut), C.POINTER(c_fp), c_fp, C.c_size_t, c_fp]
        L.nlr_prop_mlp_forward.argtypes = [c_fp] * 5 + [C.c_uint32, C.c_uint32, c_fp, c_fp]
        L.nlr_prop_mlp_backward.argtypes = [c_fp] * 6 + [C.c_uint32, C.c_uint32] + [c_fp] * 6
        L.nlr_encode_features_forward.argtypes = [C.POINTER(NlrRays), c_fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, c_fp,
                                                  C.POINTER(NlrGridDesc), C.c_uint32, c_fp, c_fp]
        L.nlr_encode_features_backward.argtypes = [C.POINTER(NlrRays), c_fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, c_fp,
                                                   C.POINTER(NlrGridDesc), C.c_uint32, c_fp, c_fp, c_fp, c_fp, c_fp]
        L.nlr_encode_features_backward_ws.argtypes = L.nlr_encode_features_backward.argtypes[:-1] + [c_fp, C.c_size_t, c_fp]
        L.nlr_train_plan_create.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_int] + [C.c_float] * 4 + [C.POINTER(c_fp), C.POINTER(C.c_uint32)]
        L.nlr_train_plan_destroy.restype = None
        L.nlr_train_plan_destroy.argtypes = [c_fp]
        L.nlr_train_act_width.restype = C.c_uint32
        L.nlr_train_act_width.argtypes = [c_fp]
        L.nlr_train_param_layout.argtypes = [c_fp, c_fp, C.c_uint32]
        L.nlr_train_pack.argtypes = [c_fp, c_fp, c_fp]
        L.nlr_mlp_train_forward.argtypes = [c_fp, c_fp, c_fp, C.c_uint32, C.c_uint32, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
        L.nlr_mlp_train_backward.argtypes = [c_fp, C.c_uint32, C.c_uint32] + [c_fp] * 11
        L.nlr_profile_begin.argtypes = [c_fp]
        L.nlr_profile_begin_kinds.argtypes = [c_fp, C.c_uint32]
        L.nlr_profile_end.argtypes = [c_fp, c_fp, c_fp, c_fp]
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    """Status -> RuntimeError with the library's message (the reference raises through TORCH_CHECK)."""
    if rc != 0:
        msg = lib().nlr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'libnerflidar_hip'} failed ({rc}): {msg}")


def ptr(t):
    """data_ptr of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
