"""ctypes binding of libvhap_hip.so (the C ABI declared in include/vhap_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing this module's
`lib()` raises -- the product path never silently runs anything but the HIP kernels.
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede the CDLL below: torch ships its own libamdhip64; loading ours first
               # would put a second HIP runtime in the process and torch.cuda would report no device

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VHAP_HIP_LIB") or os.path.join(_HERE, "lib", "libvhap_hip.so")   # (env: A/B of two builds)

c_fp = ctypes.c_void_p   # device pointers travel as integers (tensor.data_ptr())
c_i = ctypes.c_int
c_sz = ctypes.c_size_t
c_f = ctypes.c_float

# name -> (restype, [argtypes]) ; must list EVERY symbol declared in include/vhap_hip.h
SIGNATURES = {
    "vhap_abi_version": (c_i, []),
    "vhap_strerror": (ctypes.c_char_p, [c_i]),
    "vhap_stream_create": (c_i, [ctypes.POINTER(ctypes.c_void_p), c_i]),
    "vhap_stream_destroy": (c_i, [c_fp]),
    "vhap_raster_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_sz]),
    "vhap_raster_profile_offset": (c_sz, [c_i, c_i, c_i, c_i, c_sz]),
    "vhap_raster_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_sz, c_sz, c_i, c_fp]),
    "vhap_raster_interp_fwd": (c_i, [c_fp] * 5 + [c_i] * 6 + [c_fp] * 5 + [c_fp, c_sz, c_sz, c_i, c_fp]),
    "vhap_raster_shade_fwd": (c_i, [c_fp] * 7 + [c_i, c_i] + [c_fp] * 5 + [c_i] * 7 + [c_fp] * 5 + [c_fp, c_sz, c_sz, c_i, c_fp]),
    "vhap_texbin_sort_ids": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "vhap_texture_grad_binned_sorted": (c_i, [c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_sz, c_fp, c_fp]),
    "vhap_energy_total_bound": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_fp, c_fp, c_fp, c_fp]),
    "vhap_deferred_shade_bwd_work_floats": (c_sz, [c_i] * 3),
    "vhap_deferred_lights_reduce": (c_i, [c_fp] * 5 + [c_i] * 3 + [c_fp, c_fp]),
    "vhap_deferred_shade_bwd": (c_i, [c_fp] * 7 + [c_i, c_i] + [c_fp] * 11 + [c_i] * 6 + [c_fp] * 8 + [c_sz, c_fp, c_fp, c_i, c_fp]),
    "vhap_deferred_shade_bwd_list": (c_i, [c_fp] * 7 + [c_i, c_i] + [c_fp] * 11 + [c_i] * 6 + [c_fp] * 8 + [c_sz, c_fp, c_fp, c_fp, c_fp, c_i, c_fp]),
    "vhap_raster_bwd": (c_i, [c_fp] * 5 + [c_i] * 5 + [c_fp, c_fp]),
    "vhap_gbuffer_bwd": (c_i, [c_fp] * 12 + [c_i] * 5 + [c_fp] * 3),
    "vhap_interp_fwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp, c_fp, c_fp]),
    "vhap_interp_bwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp] * 4),
    "vhap_texture_num_levels": (c_i, [c_i, c_i]),
    "vhap_texture_mip_floats": (c_sz, [c_i] * 4),
    "vhap_texture_mip_build": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "vhap_texture_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "vhap_texture_bwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i] + [c_fp] * 5),
    "vhap_texture_mip_fold": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "vhap_texture_mip_fold_gather": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    "vhap_texture_grad_binned_work_bytes": (c_sz, [c_i, c_i, c_i]),
    "vhap_texture_grad_binned": (c_i, [c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "vhap_texture_grad_binned_counted": (c_i, [c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "vhap_texture_grad_binned_ids": (c_i, [c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "vhap_antialias_work_ints": (c_sz, [c_i] * 4),
    "vhap_antialias_fwd": (c_i, [c_fp] * 5 + [c_i] * 6 + [c_fp, c_fp, c_fp]),
    "vhap_antialias_bwd": (c_i, [c_fp] * 8 + [c_i] * 6 + [c_fp, c_fp, c_i, c_fp]),
    "vhap_antialias_inplace_work_ints": (c_sz, [c_i] * 4),
    "vhap_antialias_inplace_fwd": (c_i, [c_fp] * 5 + [c_i] * 5 + [c_fp, c_fp]),
    "vhap_antialias_inplace_silhouette": (c_i, [c_fp] * 3 + [c_i] * 5 + [c_fp, c_fp]),
    "vhap_antialias_inplace_pairs": (c_i, [c_fp] + [c_i] * 4 + [c_fp, c_fp]),
    "vhap_antialias_inplace_blend": (c_i, [c_fp] * 5 + [c_i] * 5 + [c_fp, c_fp]),
    "vhap_antialias_photo_bwd": (c_i, [c_fp] * 9 + [c_i] * 5 + [c_fp, c_fp, c_fp]),
    "vhap_antialias_clear_delta": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "vhap_disturb_workspace_ints": (c_sz, [c_i] * 3),
    "vhap_disturb_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "vhap_disturb_fwd_rng": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_f, c_f, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "vhap_disturb_inplace": (c_i, [c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_f, c_f, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "vhap_disturb_inplace_list": (c_i, [c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_f, c_f, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "vhap_disturb_bwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "vhap_shade_fwd": (c_i, [c_fp] * 8 + [c_i] * 4 + [c_fp] * 3 + [c_i, c_fp]),
    "vhap_shade_bwd": (c_i, [c_fp] * 9 + [c_i] * 3 + [c_fp] * 4),
    "vhap_photo_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_i, c_fp]),
    "vhap_photo_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "vhap_flame_skin_fwd": (c_i, [c_fp] * 7 + [c_i] * 6 + [c_fp] * 3 + [c_i, c_fp]),
    "vhap_flame_skin_clip_fwd": (c_i, [c_fp] * 8 + [c_i] * 6 + [c_fp] * 4 + [c_i, c_fp]),
    "vhap_flame_bwd_partial_floats": (c_sz, [c_i] * 3),
    "vhap_flame_skin_bwd": (c_i, [c_fp] * 6 + [c_i] * 5 + [c_fp] * 6 + [c_i, c_fp]),
    "vhap_verts_bwd_fused": (c_i, [c_fp] * 14 + [c_i] * 5 + [c_fp] * 8 + [c_i, c_fp]),
    "vhap_transform_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_fp, c_fp]),
    "vhap_transform_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "vhap_vnormal_fwd": (c_i, [c_fp] * 4 + [c_i, c_i, c_fp, c_fp]),
    "vhap_vnormal_bwd": (c_i, [c_fp] * 5 + [c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "vhap_vnormal_fwd_saved": (c_i, [c_fp] * 4 + [c_i, c_i, c_fp, c_fp, c_fp]),
    "vhap_vnormal_bwd_saved": (c_i, [c_fp] * 7 + [c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "vhap_frame_prep_fwd": (c_i, [c_fp] * 12 + [c_i] + [c_fp] * 3 + [c_i] * 8 + [c_fp] * 5 + [c_i, c_fp]),
    "vhap_frame_prep_bwd": (c_i, [c_fp] * 11 + [c_i] + [c_fp] * 8 + [c_i] * 8 + [c_fp] * 8 + [c_i, c_fp]),
    "vhap_frame_prep_fwd_camera": (c_i, [c_fp] * 12 + [c_i] + [c_fp] * 3 + [c_i] * 8 + [c_fp] * 5 + [c_i] +
                                   [c_fp, c_f, c_f, c_f, c_fp, c_i, c_i, c_i, c_f, c_f, c_fp, c_fp]),
    "vhap_frame_prep_bwd_camera": (c_i, [c_fp] * 11 + [c_i] + [c_fp] * 8 + [c_i] * 8 + [c_fp] * 8 + [c_i] +
                                   [c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_fp]),
    "vhap_offset_dynamic_reg": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_fp]),
    "vhap_offset_grad_finish": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "vhap_camera_fwd": (c_i, [c_fp, c_fp] + [c_i] * 5 + [c_f, c_f, c_fp, c_fp]),
    "vhap_camera_focal_fwd": (c_i, [c_fp, c_f, c_f, c_f, c_fp] + [c_i] * 4 + [c_f, c_f, c_fp, c_fp]),
    "vhap_camera_bwd": (c_i, [c_fp, c_fp] + [c_i] * 4 + [c_fp, c_fp]),
    "vhap_camera_focal_bwd": (c_i, [c_fp, c_fp] + [c_i] * 4 + [c_f, c_fp, c_fp]),
    "vhap_landmark_fwd": (c_i, [c_fp] * 5 + [c_i] * 8 + [c_f, c_i, c_i] + [c_fp] * 2 + [c_i, c_fp]),
    "vhap_landmark_bwd": (c_i, [c_fp] * 6 + [c_i] * 8 + [c_f, c_i, c_i] + [c_fp] * 3),
    "vhap_offset_reg_fwd": (c_i, [c_fp] * 8 + [c_i, c_i, c_f, c_f, c_f, c_fp, c_i, c_fp]),
    "vhap_offset_reg_bwd": (c_i, [c_fp] * 8 + [c_i, c_i, c_f, c_f, c_f, c_fp, c_fp, c_fp]),
    "vhap_offset_combine": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "vhap_offset_reg_fwd_batch": (c_i, [c_fp] * 8 + [c_i, c_i, c_i, c_f, c_f, c_f, c_fp, c_i, c_fp]),
    "vhap_offset_reg_bwd_batch": (c_i, [c_fp] * 8 + [c_i, c_i, c_i, c_f, c_f, c_f, c_fp, c_fp, c_fp]),
    "vhap_tex_prep_fwd": (c_i, [c_fp] * 3 + [c_i, c_f, c_f] + [c_fp] * 2 + [c_i, c_fp]),
    "vhap_tex_prep_mip1_fwd": (c_i, [c_fp] * 3 + [c_i, c_f, c_f] + [c_fp] * 3 + [c_i, c_fp]),
    "vhap_texture_mip_build_from": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_fp, c_i, c_fp]),
    "vhap_tex_prep_bwd": (c_i, [c_fp] * 5 + [c_i, c_fp, c_i, c_f, c_f] + [c_fp] * 2),
    "vhap_tex_prep_bwd_adam": (c_i, [c_fp] * 5 + [c_i, c_fp, c_i, c_f, c_f] + [c_fp] * 5 + [c_f, c_f, c_f, c_i, c_fp]),
    "vhap_tex_prep_bwd_base": (c_i, [c_fp] * 5 + [c_i, c_fp, c_i, c_f, c_f] + [c_fp] * 3),
    "vhap_tex_prep_bwd_adam_base": (c_i, [c_fp] * 5 + [c_i, c_fp, c_i, c_f, c_f] + [c_fp] * 5 + [c_f, c_f, c_f, c_fp, c_i, c_fp]),
    "vhap_tex_pca_fwd": (c_i, [c_fp] * 3 + [c_i, c_i, c_i, c_f] + [c_fp] * 4),
    "vhap_tex_pca_bwd": (c_i, [c_fp] * 4 + [c_i, c_i, c_i, c_f] + [c_fp] * 4),
    "vhap_tex_prep_bwd_adam_rows": (c_i, [c_fp] * 5 + [c_i, c_i, c_i, c_f, c_f] + [c_fp] * 5 + [c_f, c_f, c_f, c_i, c_fp]),
    "vhap_tex_carry_halo_floats": (c_sz, [c_i]),
    "vhap_tex_carry_prime": (c_i, [c_fp] * 3 + [c_i, c_f, c_f] + [c_fp] * 5),
    "vhap_tex_carry_border": (c_i, [c_i, c_f, c_fp, c_fp, c_fp, c_i, c_fp]),
    "vhap_tex_finish_carry": (c_i, [c_fp] * 6 + [c_i, c_fp, c_i, c_f, c_f] + [c_fp] * 5 + [c_f, c_f, c_f] + [c_fp] * 3 + [c_i, c_fp]),
    "vhap_energy_finalize": (c_i, [c_fp] * 5 + [c_f, c_f, c_i, c_i, c_i, c_fp, c_fp]),
    "vhap_energy_total": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_fp, c_fp]),
    "vhap_sum_frames": (c_i, [c_fp, c_i, c_i, c_fp, c_fp]),
    "vhap_adam_step": (c_i, [c_i] + [c_fp] * 8 + [c_f, c_f, c_f, c_i, c_fp]),
    "vhap_adam_advance": (c_i, [c_fp, c_fp]),
    "vhap_set_floats": (c_i, [c_fp, ctypes.POINTER(c_f), c_i, c_fp]),
    "vhap_raster_bin_vnormal": (c_i, [c_fp, c_fp, c_fp] + [c_i] * 5 + [c_fp, c_sz, c_sz, c_i] + [c_fp] * 6),
    "vhap_raster_bin_vnormal_prefill": (c_i, [c_fp, c_fp, c_fp] + [c_i] * 5 + [c_fp, c_sz, c_sz, c_i] + [c_fp] * 5 + [c_fp, c_fp, c_fp, c_i] + [c_fp] * 5),
    "vhap_raster_shade_stats": (c_i, [c_i] * 4 + [c_fp, c_sz, c_sz, c_i, c_fp, c_fp]),
    "vhap_photo_fwd_total": (c_i, [c_fp, c_fp, c_i, c_i, c_i] + [c_fp] * 6 + [c_f] * 3 + [c_fp] * 4 + [c_fp, c_fp, c_i, c_fp]),
    "vhap_frame_ingest": (c_i, [c_fp, c_fp, c_fp] + [c_i] * 5 + [c_fp] * 4),
    "vhap_frame_color_correct": (c_i, [c_fp, c_fp, c_fp] + [c_i] * 4 + [c_fp, c_fp]),
    "vhap_frame_resize_u8": (c_i, [c_fp] + [c_i] * 4 + [c_fp, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_fp]),
    "vhap_batch_feed": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i] + [c_fp, c_fp, c_i] * 3 + [c_fp, c_fp, c_fp]),
    "vhap_conv2d_nhwc": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i] + [c_fp] * 4 + [c_i] * 4 + [c_fp, c_i, c_i, c_i, c_fp]),
    "vhap_conv2d_nhwc_ws": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i] + [c_fp] * 4 + [c_i] * 4 + [c_fp, c_i, c_i, c_fp, ctypes.c_longlong, c_i, c_fp]),
    "vhap_nhwc_avgpool2": (c_i, [c_fp] + [c_i] * 4 + [c_fp, c_fp]),
    "vhap_nhwc_upsample2_add": (c_i, [c_fp, c_fp] + [c_i] * 4 + [c_fp, c_fp]),
    "vhap_nhwc_maxpool2": (c_i, [c_fp] + [c_i] * 4 + [c_fp, c_fp]),
    "vhap_nhwc_l2norm": (c_i, [c_fp, ctypes.c_longlong, c_i, c_fp, ctypes.c_float, c_fp, c_fp]),
    "vhap_nhwc_add": (c_i, [c_fp, c_fp, c_fp, ctypes.c_longlong, c_fp, c_fp]),
    "vhap_plan_from_graph": (c_i, [c_fp, c_i, ctypes.POINTER(ctypes.c_void_p)]),
    "vhap_plan_destroy": (c_i, [c_fp]),
    "vhap_plan_set_side_base": (c_i, [c_i]),
    "vhap_plan_touch_side_streams": (c_i, [c_i, c_fp]),
    "vhap_plan_side_stream_wait": (c_i, [c_i, c_fp]),
    "vhap_plan_pool_release": (c_i, []),
    "vhap_plan_info": (c_i, [c_fp] + [ctypes.POINTER(c_i)] * 3),
    "vhap_plan_describe": (c_sz, [c_fp, ctypes.c_char_p, c_sz]),
    "vhap_plan_node_name": (c_i, [c_fp, c_i, ctypes.c_char_p, c_sz]),
    "vhap_plan_launch": (c_i, [c_fp, c_fp, c_i]),
    "vhap_plan_join": (c_i, [c_fp, c_fp]),
    "vhap_plan_open_tails": (c_i, [c_fp, ctypes.POINTER(c_i), c_i]),
    "vhap_plan_free_heads": (c_i, [c_fp, ctypes.POINTER(c_i), c_i]),
    "vhap_plan_node_handle": (c_fp, [c_fp, c_i]),
    "vhap_capture_nodes": (c_i, [c_fp, ctypes.POINTER(ctypes.c_void_p), c_i]),
    "vhap_plan_launch_timed": (c_i, [c_fp, c_fp, ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_i]),
}

ABI_VERSION = 10
# call_flags of include/vhap_hip.h (per-call arguments since ABI 2; the library keeps no mutable state)
CALL_ACC_PREZEROED, CALL_AA_PASSTHROUGH_DONE, CALL_ADAM_KEEP_STEP, CALL_ADAM_STEP_ADVANCED, CALL_OFFSET_PER_FRAME = 1, 2, 4, 16, 32
CALL_PLAN_DEFER_JOIN = 64
CALL_DELTA_UNSCALED = 128
CALL_TEX_TERMS_CONSUME = 256
CALL_SKIP_BG_GRAD = 512
CONV_IN_RELU, CONV_OUT_RELU, CONV_ACCUMULATE = 1, 2, 4

_lib = None


class VhapHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise VhapHipError(
                f"{SO_PATH} not found: build it with `python -m vhap_amd.build` "
                "(there is no CPU / eager fallback for the hot path)")
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.vhap_abi_version() != ABI_VERSION:
            raise VhapHipError("libvhap_hip.so ABI version mismatch")
        _lib = L
    return _lib


def debug_set_flags(flags):
    """Profiling-only A/B switches of the CALLING THREAD (thread-local in the library; not part of the stable ABI)."""
    fn = lib().vhap_debug_set_flags
    fn.restype, fn.argtypes = None, [c_i]
    fn(int(flags))


def check(code, what=""):
    if code != 0:
        msg = lib().vhap_strerror(code).decode()
        raise VhapHipError(f"{what}: {msg} ({code})")
    if ACCESS is not None:
        ACCESS.close(what)


# ---- which buffers does a captured node touch? -------------------------------------------------------------------------------------
# The deferred join of a step plan (tracker.GraphedStep) lets the next replay's head run under the previous replay's open tail: safe iff
# the two touch disjoint memory.  That is decided on the BUFFERS, not on kernel names: while a step is being captured, every device
# pointer handed to a C-ABI call is recorded (ops._p -> AccessLog.touch) together with the number of nodes the capture held before and
# after the call (vhap_capture_node_count), so that every node of the plan maps back to the call that created it and to the byte ranges
# that call was given.  Nodes no recorded call accounts for (the host framework's own fills / copies inside the capture) are UNKNOWN: a
# plan with an unknown node among its open tails or free heads does not defer.
ACCESS = None


class AccessLog:
    CAP = 4096

    def __init__(self):
        self.owner = {}               # graph node handle -> (call name, [(address, bytes), ...]) | None (a node no recorded call created)
        self._cur = None
        self._buf = (ctypes.c_void_p * self.CAP)()

    def _new_nodes(self):
        """handles of the nodes the capture has gained since the last look"""
        n = int(lib().vhap_capture_nodes(torch.cuda.current_stream().cuda_stream, self._buf, self.CAP))
        return [h for h in (self._buf[i] for i in range(min(n, self.CAP))) if h not in self.owner]

    def touch(self, t):
        if self._cur is None:         # first pointer of a call: whatever appeared since the last call closed was not created by a recorded call
            for h in self._new_nodes():
                self.owner[h] = None
            self._cur = []
        if t.numel():
            if t.is_contiguous():
                self._cur.append((t.data_ptr(), t.numel() * t.element_size()))
            else:                     # a strided view: numel * element_size under-reports what a kernel may touch -- its whole storage, conservatively
                st = t.untyped_storage()
                self._cur.append((st.data_ptr(), st.nbytes()))

    def close(self, what):
        # a call that handed over no recorded pointer (raw data_ptr() arguments, or none at all) cannot vouch for the nodes that appeared
        # since the last look: they stay UNKNOWN (None) -- a plan with an unknown node among its open tails or free heads does not defer
        owner = (what, self._cur) if self._cur is not None else None
        for h in self._new_nodes():
            self.owner[h] = owner
        self._cur = None

    def ranges_of_node(self, handle):
        """(call name, byte ranges) of the call that created the graph node `handle` (None: no recorded call did)"""
        return self.owner.get(handle)

    def __enter__(self):
        global ACCESS
        self._prev, ACCESS = ACCESS, self
        return self

    def __exit__(self, *a):
        global ACCESS
        ACCESS = self._prev
        return False


def ranges_overlap(a, b):
    """first overlapping pair of two lists of (address, bytes), or None"""
    for pa, na in a:
        for pb, nb in b:
            if pa < pb + nb and pb < pa + na:
                return (pa, na), (pb, nb)
    return None


# ---- streams of our own ----------------------------------------------------------------------------------------------------------
# torch.cuda.Stream() does not create a stream: it hands out the next of 32 pooled streams per device and priority, round-robin, so the
# 33rd Stream object of a process IS the first one again.  The step executor (vhap_amd/step.py, tracker.GraphedStep) forks work onto
# side streams inside a stream capture and launches the captured graph on yet another stream; a process that had created enough streams
# before (a test suite, a long-running tracker) got a launch stream that aliased one of the captured side streams, and hipGraphLaunch of
# ROCm 7 dereferenced a null pointer setting up the graph's parallel branches.  The library therefore owns its streams
# (vhap_stream_create), one per (thread, device, role), created on first use and kept for the life of the process.
import threading

_streams = threading.local()


def private_stream(role, device=None, high_priority=False):
    """The calling thread's stream for `role` ('launch', 'side', 'side2', 'warm', 'tex', ...) on `device`: a torch.cuda.ExternalStream
    over a HIP stream this library created -- never a member of torch's stream pool, so it aliases no other stream of the process."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    table = getattr(_streams, "table", None)
    if table is None:
        table = _streams.table = {}
    key = (dev.index, role, bool(high_priority))
    st = table.get(key)
    if st is None:
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            check(lib().vhap_stream_create(ctypes.byref(handle), 1 if high_priority else 0), "vhap_stream_create")
        st = table[key] = torch.cuda.ExternalStream(handle.value, device=dev)
    return st
