"""ctypes binding of libdvae_hip.so (the C ABI declared in include/dvae.h).

PyTorch is used only for device memory and streams: every call takes raw
``data_ptr()`` addresses and enqueues on ``torch.cuda.current_stream()``.
There is NO fallback: if the library is missing or reports an error, a
RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVAE_LIB") or os.path.join(_HERE, "libdvae_hip.so")      # DVAE_LIB: diagnostic builds (tools/exp_rows2.sh)

ACT_NONE, ACT_TANH, ACT_RELU, ACT_SIGMOID, ACT_EXP = 0, 1, 2, 3, 4
ABI_VERSION = 1

_lib = None

c_vp, c_i, c_i64, c_f, c_d, c_sz = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                    ctypes.c_double, ctypes.c_size_t)

# name -> (restype, argtypes); mirrors include/dvae.h, include/dvae_train.h and include/dvae_mcem.h
SIGNATURES = {
    "dvae_build_has_diag": (c_i, []),
    "dvae_abi_version": (c_i, []),
    "dvae_last_error": (ctypes.c_char_p, []),
    "dvae_device_count": (c_i, []),
    "dvae_linear_act_fwd": (c_i, [c_vp, c_i, c_i, c_vp, c_i, c_i, c_vp, c_i, c_vp, c_vp, c_i, c_i64, c_i, c_i, c_vp]),
    "dvae_act_bwd": (c_i, [c_vp, c_i, c_vp, c_i, c_vp, c_i, c_i64, c_i, c_i, c_vp]),
    "dvae_linear_bwd_data": (c_i, [c_vp, c_i, c_vp, c_i, c_i, c_vp, c_i, c_i64, c_i, c_i, c_i, c_vp]),
    "dvae_linear_bwd_weight": (c_i, [c_vp, c_i, c_vp, c_i, c_i, c_vp, c_i, c_i, c_vp, c_i, c_vp, c_i64, c_i, c_i, c_vp]),
    "dvae_linear_bwd_weight_det": (c_i, [c_vp, c_i, c_vp, c_i, c_i, c_vp, c_i, c_i, c_vp, c_i, c_vp, c_i64, c_i, c_i, c_vp, c_vp]),
    "dvae_linear_bwd_weight_workspace_bytes": (ctypes.c_size_t, [c_i64, c_i, c_i, c_i]),
    "dvae_reparam_fwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dvae_reparam_bwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dvae_elbo_workspace_bytes": (c_sz, [c_i64]),
    "dvae_elbo_fwd": (c_i, [c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_f, c_i64, c_i, c_i, c_vp, c_vp, c_vp, c_vp]),
    "dvae_elbo_bwd": (c_i, [c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_vp, c_i64, c_i, c_i, c_vp, c_i, c_vp, c_vp, c_vp]),
    "dvae_elbo_bwd3": (c_i, [c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i, c_i, c_vp, c_i, c_vp, c_vp, c_vp]),
    "dvae_bce_fwd": (c_i, [c_vp, c_vp, c_f, c_i64, c_i, c_i, c_vp, c_vp, c_vp]),
    "dvae_bce_bwd": (c_i, [c_vp, c_vp, c_f, c_vp, c_i64, c_i, c_i, c_vp, c_vp, c_vp]),
    "dvae_isrows_fwd": (c_i, [c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_f, c_i64, c_i, c_i, c_vp, c_vp, c_vp]),
    "dvae_isrows_bwd": (c_i, [c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_i64, c_i, c_i, c_vp, c_i, c_vp, c_vp, c_vp]),
    "dvae_bce2_fwd": (c_i, [c_vp, c_vp, c_vp, c_f, c_i64, c_i, c_vp, c_vp, c_vp]),
    "dvae_bce2_bwd": (c_i, [c_vp, c_vp, c_vp, c_f, c_vp, c_i64, c_i, c_vp, c_vp, c_vp, c_vp]),
    "dvae_sqerr_fwd": (c_i, [c_i, c_vp, c_vp, c_vp, c_i64, c_i, c_vp, c_vp, c_vp]),
    "dvae_sqerr_bwd": (c_i, [c_i, c_vp, c_vp, c_vp, c_vp, c_i64, c_i, c_vp, c_vp, c_vp, c_vp]),
    "dvae_adam_step": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i64, c_d, c_d, c_d, c_d, c_i, c_d, c_vp]),
    "dvae_stft": (c_i, [c_vp, c_i, c_i64, c_vp, c_i, c_i, c_i64, c_vp, c_i, c_vp]),
    "dvae_stft_f32": (c_i, [c_vp, c_i64, c_vp, c_i, c_i, c_i64, c_vp, c_i, c_vp]),
    "dvae_istft_workspace_bytes": (c_sz, [c_i64, c_i]),
    "dvae_istft_workspace_bytes_hop": (c_sz, [c_i64, c_i, c_i]),
    "dvae_istft": (c_i, [c_vp, c_i64, c_i64, c_vp, c_i, c_i, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dvae_istft_frames": (c_i, [c_vp, c_i64, c_i64, c_vp, c_i, c_i, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dvae_istft_f32": (c_i, [c_vp, c_i64, c_i64, c_i, c_vp, c_i, c_i, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dvae_transpose": (c_i, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "dvae_gather_rows": (c_i, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i, c_vp, c_i64, c_vp, c_vp]),
    "dvae_vad_workspace_bytes": (c_sz, [c_i64]),
    "dvae_vad_labels": (c_i, [c_vp, c_i, c_i64, c_i, c_i, c_i64, c_d, c_vp, c_vp, c_vp]),
    "dvae_ibm_workspace_bytes": (c_sz, []),
    "dvae_ibm_labels": (c_i, [c_vp, c_i64, c_i64, c_f, c_f, c_vp, c_vp, c_vp, c_vp]),
    # include/dvae_train.h (plan pointers are passed with ctypes.byref)
    "dvae_train_plan": (c_i, [c_i, c_i, c_i, c_i64, c_i, c_vp]),
    "dvae_train_init": (c_i, [c_vp, c_vp, c_vp, c_vp]),
    "dvae_train_repack": (c_i, [c_vp, c_vp, c_vp, c_vp]),
    "dvae_train_grads": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_f, c_i, c_vp]),
    "dvae_train_grads_group": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_f, c_i, c_i, c_vp]),
    "dvae_train_group_range": (c_i, [c_vp, c_i, c_vp, c_vp, c_vp]),
    "dvae_train_apply": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_d, c_d, c_d, c_d, c_d, c_vp, c_vp]),
    "dvae_train_step": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_f, c_i, c_d, c_d, c_d, c_d, c_vp, c_vp]),
    "dvae_train_step_deferred": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_f, c_i, c_d, c_d, c_d, c_d, c_vp, c_vp]),
    "dvae_train_flush": (c_i, [c_vp, c_vp, c_vp]),
    "dvae_train_pending": (c_i, [c_vp]),
    "dvae_train_can_defer": (c_i, [c_vp, c_vp]),
    "dvae_train_eval": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_f, c_vp, c_vp]),
    "dvae_train_noise": (c_i, [c_vp, ctypes.c_uint64, c_vp, c_vp]),
    "dvae_module_forward": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_i, c_vp]),
    "dvae_module_backward": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_vp]),
    "dvae_train_profile": (c_i, [c_i]),
    "dvae_train_debug_stamps": (c_i, [c_vp]),
    "dvae_mcem_debug_stamps": (c_i, [c_vp]),
    "dvae_train_profile_read": (c_i, [c_vp, c_vp]),
    "dvae_comm_create": (c_i, [c_i, c_i, c_i64, c_vp, c_vp]),
    "dvae_comm_connect": (c_i, [c_vp, c_vp]),
    "dvae_allreduce_flat": (c_i, [c_vp, c_vp, c_i, c_i64, c_vp, c_vp]),
    "dvae_comm_set_timeout_ms": (c_i, [c_vp, c_i64]),
    "dvae_comm_status": (c_i, [c_vp, c_vp]),
    "dvae_comm_destroy": (c_i, [c_vp]),
    # include/dvae_mcem.h
    "dvae_mcem_plan": (c_i, [c_i, c_i, c_vp]),
    "dvae_mcem_pack": (c_i, [c_vp, c_vp, c_i, c_vp, c_vp, c_i, c_vp, c_vp, c_i, c_vp, c_vp, c_vp]),
    "dvae_mcem_sample": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_f, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dvae_mcem_decode": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i64, c_vp, c_vp]),
    "dvae_mcem_m_step_workspace_bytes": (c_sz, [c_i64, c_i, c_i]),
    "dvae_mcem_m_step_batch": (c_i, [c_vp, c_vp, c_i, c_i64, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dvae_mcem_m_step": (c_i, [c_vp, c_vp, c_i, c_i64, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dvae_mcem_em_iteration": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_f, c_i64, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dvae_mcem_em_iteration_lazy": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_f, c_i64, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dvae_mcem_cost_flush": (c_i, [c_i, c_i64, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dvae_mcem_wiener": (c_i, [c_vp, c_i, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
}


def load(path=None):
    """Load the shared library and type every exported symbol (no GPU needed)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"libdvae_hip.so not found at {p}: build it with `python disentangled-vae_amd/build.py` "
            "(there is no CPU fallback for CUDA tensors)")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.dvae_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libdvae_hip.so ABI {lib.dvae_abi_version()} != binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def is_built():
    return os.path.exists(LIB_PATH)


def check(rc, what):
    if rc != 0:
        msg = load().dvae_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current HIP stream of the current device as a void* (the raw-handle query: building a torch.cuda.Stream object per
    launch costs ~6 us, four times per step of the module path)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def as_f32_2d(t, what):
    """Contiguous-rows fp32 CUDA matrix view (leading dims flattened)."""
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a CUDA tensor")
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: the HIP path computes in float32, got {t.dtype}")
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(-1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < t2.shape[1]):
        t2 = t2.contiguous()
    return t2


def ld(t2):
    return t2.stride(0) if t2.shape[0] > 1 else max(t2.shape[1], 1)
