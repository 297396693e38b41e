"""ctypes binding of the C ABI in include/siggan.h.

The shared library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile`` and must be
present: there is no CPU or PyTorch fallback for this path -- a missing or stale library raises.
torch is imported first so the library resolves HIP against the runtime torch already loaded
(one HIP runtime per process)."""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the library; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIGGAN_LIB_PATH: load another build of the same ABI (A/B measurements of kernel variants); it must exist -- a missing
# library raises either way, there is nothing to fall back to
LIB_PATH = os.environ.get("SIGGAN_LIB_PATH") or os.path.join(_HERE, "libsiggan_hip.so")
ABI_VERSION = 3
M_COUNT = 16
METRIC_INDEX = {"d_loss": 0, "d_loss_real": 1, "d_loss_fake": 2, "d_real_mean": 3, "d_fake_mean": 4,
                "d_real_acc": 5, "d_fake_acc": 6, "d_grad_norm": 7, "g_loss": 8, "g_fake_mean": 9,
                "g_grad_norm": 10, "d_skipped": 11, "g_skipped": 12}


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("latent_dim", C.c_int32), ("image_size", C.c_int32),
                ("image_channels", C.c_int32), ("max_batch", C.c_int32), ("dropout", C.c_float),
                ("leaky_slope", C.c_float), ("seed", C.c_uint64), ("dtype", C.c_int32), ("f16_grad_scale", C.c_float),
                ("spectral_norm", C.c_int32)]


DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f16": 2, "fp16": 2, "float16": 2}


class Storage(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "g_params", "g_grads", "g_exp_avg", "g_exp_avg_sq", "g_adam_steps", "g_bn_running_mean",
        "g_bn_running_var", "g_bn_batches", "d_params", "d_grads", "d_exp_avg", "d_exp_avg_sq", "d_adam_steps",
        "d_sn_u", "d_sn_v")]


class Hyper(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("label_smoothing", C.c_float), ("clip_max_norm", C.c_float), ("grad_scale", C.c_float)]


class MlpConfig(C.Structure):       # include/siggan_mlp.h
    _fields_ = [("device", C.c_int32), ("latent_dim", C.c_int32), ("image_size", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * 4), ("max_batch", C.c_int32), ("leaky_slope", C.c_float), ("seed", C.c_uint64)]


class MlpStorage(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "g_params", "g_grads", "g_exp_avg", "g_exp_avg_sq", "g_adam_steps", "g_bn_running_mean",
        "g_bn_running_var", "g_bn_batches", "d_params", "d_grads", "d_exp_avg", "d_exp_avg_sq", "d_adam_steps")]


_P, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
_SIGNATURES = {
    "siggan_abi_version": (C.c_int, []),
    "siggan_last_error": (C.c_char_p, []),
    "siggan_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "siggan_destroy": (C.c_int, [_P]),
    "siggan_param_count": (_I64, [_P, C.c_int]),
    "siggan_param_tensors": (_I32, [_P, C.c_int]),
    "siggan_param_span": (C.c_int, [_P, C.c_int, _I32, C.POINTER(_I64), C.POINTER(_I64)]),
    "siggan_bn_count": (_I64, [_P]),
    "siggan_bn_layers": (_I32, [_P]),
    "siggan_sn_count": (_I64, [_P, C.c_int]),
    "siggan_workspace_bytes": (_I64, [_P]),
    "siggan_bind": (C.c_int, [_P, C.POINTER(Storage)]),
    "siggan_params_changed": (C.c_int, [_P]),
    "siggan_seed": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "siggan_rng_state": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "siggan_set_mode": (C.c_int, [_P, _I32]),
    "siggan_set_step_variant": (C.c_int, [_P, _I32]),
    "siggan_g_forward": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "siggan_d_forward": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P, _P]),
    "siggan_d_step": (C.c_int, [_P, _P, _I32, _P, _P, C.POINTER(Hyper), _P, _P, _P]),
    "siggan_g_step": (C.c_int, [_P, _I32, _P, C.POINTER(Hyper), _P, _P, _P]),
    "siggan_d_grads": (C.c_int, [_P, _P, _I32, _P, _P, C.POINTER(Hyper), _P, _P]),
    "siggan_step_begin": (C.c_int, [_P, _P, _I32, _P, _P, _P, C.POINTER(Hyper), _P, _P]),
    "siggan_stage_real": (C.c_int, [_P, _P, _I32, _P]),
    "siggan_d_apply": (C.c_int, [_P, C.POINTER(Hyper), _P, _P, _P]),
    "siggan_g_grads": (C.c_int, [_P, _I32, _P, C.POINTER(Hyper), _P, _P]),
    "siggan_g_apply": (C.c_int, [_P, C.POINTER(Hyper), _P, _P, _P]),
    "siggan_op_conv4x4s2": (C.c_int, [_P, _I32, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "siggan_op_conv4x4s2_wgrad": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "siggan_op_adam": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, C.POINTER(Hyper), _P]),
    "siggan_op_randn": (C.c_int, [_P, _P, _I64, _P]),
    "siggan_augment_batch": (C.c_int, [_I32, _P, _I64, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "siggan_comm_unique_id": (C.c_int, [_P]),
    "siggan_comm_init": (C.c_int, [_P, _I32, _I32, _P]),
    "siggan_comm_destroy": (C.c_int, [_P]),
    "siggan_comm_world": (_I32, [_P]),
    "siggan_comm_broadcast": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "siggan_device_info": (C.c_int, [_I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I64)]),
    "siggan_prof_enable": (C.c_int, [_P, _I32]),
    "siggan_prof_slots": (_I32, []),
    "siggan_prof_read": (C.c_int, [_P, _I32, C.c_char_p, _I32, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "siggan_debug_tensor": (C.c_int, [_P, C.c_char_p, _I32, _P, _I64, _P]),
    # fully-connected extension (include/siggan_mlp.h; build-defined, parity unpinned)
    "mlpgan_create": (C.c_int, [C.POINTER(MlpConfig), C.POINTER(_P)]),
    "mlpgan_destroy": (C.c_int, [_P]),
    "mlpgan_param_count": (_I64, [_P, C.c_int]),
    "mlpgan_param_tensors": (_I32, [_P, C.c_int]),
    "mlpgan_bn_count": (_I64, [_P]),
    "mlpgan_bind": (C.c_int, [_P, C.POINTER(MlpStorage)]),
    "mlpgan_seed": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "mlpgan_g_forward": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "mlpgan_d_forward": (C.c_int, [_P, _P, _I32, _P, _P]),
    "mlpgan_d_step": (C.c_int, [_P, _P, _I32, _P, C.POINTER(Hyper), _P, _P]),
    "mlpgan_g_step": (C.c_int, [_P, _I32, _P, C.POINTER(Hyper), _P, _P]),
    "mlpgan_op_gemm": (C.c_int, [_I32, _I32, _P, _P, _P, _I32, _I32, _I32, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C signature-gan_amd/csrc`). This path has no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export the symbol
        fn.restype, fn.argtypes = res, args
    if lib.siggan_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libsiggan_hip.so ABI {lib.siggan_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


class SigganError(RuntimeError):
    pass


def check(rc):
    """Map a C return code to the exception the reference would raise for the same misuse."""
    if rc == 0:
        return
    msg = load().siggan_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)           # e.g. bad image size: generator_vanilla_gan.py:106-107
    raise SigganError(f"siggan error {rc}: {msg}")
