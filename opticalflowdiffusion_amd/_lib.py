"""ctypes binding of libofd_hip.so (the C-ABI in include/ofd.h).

There is no fallback: if the library is missing, or a tensor is not on the GPU, the call
raises.  PyTorch only provides device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFD_LIB") or os.path.join(_HERE, "lib", "libofd_hip.so")      # OFD_LIB: A/B runs of alternative builds

c_void_p, c_int, c_size_t, c_float, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float, ctypes.c_char_p


class OfdError(RuntimeError):
    pass


class UnetConfig(ctypes.Structure):
    _fields_ = [("dim", c_int), ("channels", c_int), ("out_dim", c_int), ("eps_mode", c_int), ("no_time", c_int)]


class ConvSrc(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("channels", c_int), ("src_channels", c_int), ("ch_offset", c_int),
                ("upsample", c_int), ("unshuffle", c_int), ("p1", c_int), ("p2", c_int)]


class ConvArgs(ctypes.Structure):
    _fields_ = [("B", c_int), ("H", c_int), ("W", c_int), ("ksize", c_int), ("n_src", c_int),
                ("src", ConvSrc * 4), ("Cout", c_int), ("weight", c_void_p), ("bias", c_void_p),
                ("in_scale", c_void_p), ("in_shift", c_void_p), ("residual", c_void_p), ("res_act", c_void_p),
                ("res_scale", c_void_p), ("res_shift", c_void_p), ("out", c_void_p), ("gn_partial", c_void_p),
                ("out2", c_void_p), ("residual2", c_void_p), ("split", c_int), ("up2_phase", c_int)]


# name -> (restype, argtypes); every symbol declared in include/ofd.h
# host callback of ofd_unet_backward: (begin, end, user) float range of the flat gradient buffer
GRAD_READY = ctypes.CFUNCTYPE(None, c_size_t, c_size_t, c_void_p)

SIGNATURES = {
    "ofd_version": (c_int, []),
    "ofd_last_error": (c_char_p, []),
    "ofd_splat_workspace_bytes": (c_size_t, [c_int] * 3),
    "ofd_splat_fwd": (c_int, [c_void_p] * 3 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "ofd_splat_corners": (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_void_p]),
    "ofd_splat_bwd_in": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    "ofd_splat_bwd_flow": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "ofd_warp_prep": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "ofd_augment": (c_int, [c_void_p] * 8 + [c_int] * 4 + [c_void_p]),
    "ofd_augment_table": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ofd_batch_stats_ws_doubles": (c_size_t, []),
    "ofd_batch_stats": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p]),
    "ofd_warp_holes": (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_void_p]),
    "ofd_grid_warp_fwd": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "ofd_splat_pyramid_workspace_bytes": (c_size_t, [c_int] * 4),
    "ofd_splat_pyramid_fwd": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "ofd_splat_pyramid_bwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "ofd_pyramid_charbonnier_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "ofd_pyramid_charbonnier_bwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "ofd_grid_warp_bwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "ofd_grid_warp_corners": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "ofd_q_sample": (c_int, [c_void_p] * 5 + [c_int, c_size_t, c_void_p]),
    "ofd_ddpm_update": (c_int, [c_void_p] * 8 + [c_int, c_size_t, c_void_p]),
    "ofd_ddim_update": (c_int, [c_void_p] * 8 + [c_int] + [c_void_p] * 2 + [c_int, c_size_t, c_void_p]),
    "ofd_nan_mse_sum": (c_int, [c_void_p] * 2 + [c_size_t, c_void_p, c_void_p]),
    "ofd_nan_mse_result_doubles": (c_size_t, []),
    "ofd_adam_chunk": (c_int, []),
    "ofd_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_float] * 6 + [c_int, c_void_p]),
    "ofd_unet_create": (c_int, [ctypes.POINTER(UnetConfig), ctypes.POINTER(c_void_p)]),
    "ofd_unet_destroy": (None, [c_void_p]),
    "ofd_unet_num_params": (c_int, [c_void_p]),
    "ofd_unet_param_name": (c_char_p, [c_void_p, c_int]),
    "ofd_unet_param_numel": (c_size_t, [c_void_p, c_int]),
    "ofd_unet_param_shape": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int)]),
    "ofd_unet_set_param": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "ofd_unet_bind_param_buffer": (c_int, [c_void_p, c_void_p, c_size_t]),
    "ofd_unet_prepare": (c_int, [c_void_p, c_void_p]),
    "ofd_unet_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "ofd_unet_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ofd_unet_set_graph": (c_int, [c_void_p, c_int]),
    "ofd_unet_set_split_streams": (c_int, [c_void_p, c_int, c_int]),
    "ofd_unet_read_tap": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t, c_void_p]),
    "ofd_unet_set_debug_taps": (c_int, [c_void_p, c_int]),
    "ofd_unet_set_profiling": (c_int, [c_void_p, c_int]),
    "ofd_unet_set_deterministic": (c_int, [c_void_p, c_int]),
    "ofd_unet_deterministic_misses": (ctypes.c_long, [c_void_p]),
    "ofd_unet_prof_count": (c_int, [c_void_p]),
    "ofd_unet_prof_name": (c_char_p, [c_void_p, c_int]),
    "ofd_unet_prof_read": (c_int, [c_void_p, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong),
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ofd_unet_prof_reset": (c_int, [c_void_p]),
    "ofd_unet_prof_dump_path": (c_int, [c_void_p, c_char_p]),
    "ofd_conv_forward": (c_int, [ctypes.POINTER(ConvArgs), c_void_p]),
    "ofd_conv_forward_pool2": (c_int, [ctypes.POINTER(ConvArgs), c_void_p]),
    "ofd_conv_gn_partial_count": (c_size_t, [c_int] * 4),
    "ofd_conv_weight_elems": (c_size_t, [c_int] * 3),
    "ofd_conv_dgrad_weight_prep": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ofd_conv_wgrad": (c_int, [ctypes.POINTER(ConvArgs), c_void_p, c_void_p, c_void_p]),
    "ofd_conv7_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ofd_conv_wgrad_finish": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_float, c_int, c_int, c_void_p]),
    "ofd_grad_scatter": (c_int, [c_void_p, c_int, c_int, c_void_p] + [c_int] * 8 + [c_void_p]),
    "ofd_channel_sum": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ofd_gn_bwd_workspace_floats": (c_size_t, [c_int] * 4),
    "ofd_gn_silu_backward": (c_int, [c_void_p] * 8 + [c_int, c_int] + [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "ofd_affine_silu": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "ofd_layernorm_c_backward": (c_int, [c_void_p] * 5 + [c_size_t, c_int, c_float, c_int, c_void_p]),
    "ofd_final_conv_backward": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "ofd_la_workspace_floats": (c_size_t, [c_int, c_int]),
    "ofd_la_bwd_workspace_floats": (c_size_t, [c_int, c_int]),
    "ofd_linear_attention_core": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "ofd_linear_attention_core_backward": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "ofd_flash_attention": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p]),
    "ofd_flash_attention_backward": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "ofd_unet_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "ofd_unet_param_floats": (c_size_t, [c_void_p]),
    "ofd_unet_param_offset": (c_size_t, [c_void_p, c_int]),
    "ofd_unet_bind_grad_buffer": (c_int, [c_void_p, c_void_p, c_size_t]),
    "ofd_unet_train_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ofd_unet_backward": (c_int, [c_void_p, c_void_p, GRAD_READY, c_void_p, c_void_p]),
    "ofd_nan_mse_grad": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ofd_layernorm_c": (c_int, [c_void_p] * 4 + [c_size_t, c_int, c_float, c_void_p]),
    "ofd_time_mlp": (c_int, [c_void_p] * 7 + [c_int, c_int, c_void_p]),
    "ofd_gn_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ofd_conv_upsample_phase_weight_prep": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ofd_conv_weight_prep": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_float, c_int, c_void_p]),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built (python -m opticalflowdiffusion_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfdError(f"{LIB_PATH} is missing: build it with `python -m opticalflowdiffusion_amd.build` "
                           "(there is no CPU or PyTorch fallback for this path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise OfdError(f"libofd_hip error {rc}: {lib().ofd_last_error().decode()}")


def last_error():
    return lib().ofd_last_error().decode()


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise OfdError("opticalflowdiffusion_amd runs on the GPU only: got a CPU tensor "
                           "(the reference asserts the same, softsplat_new.py:444)")


def f32c(t):
    """contiguous fp32 view/copy of a GPU tensor"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
