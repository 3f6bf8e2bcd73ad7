"""ctypes binding of libwindgnn_hip.so (the C ABI declared in include/windgnn.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is
missing or a call fails this module raises, loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WGNN_LIB: another build of the SAME library (same-box A/B timing of kernel variants, tools/exp); never a different backend
LIB_PATH = os.environ.get("WGNN_LIB") or os.path.join(_HERE, "csrc", "libwindgnn_hip.so")

MATH_F32 = 0
MATH_F16X3 = 1
MATH_F16 = 2
MATH_F16X3G = 3       # f16x3 whose backward gate gradients travel as one fp16 plane from B*T >= 4096 (include/windgnn.h)
ADJ_DENSE = 0
ADJ_CSR = 1
IO_F32, IO_F16, IO_BF16 = 0, 1, 2   # wgnn_io: element type of X, Y and the labels
STATUS_BYTES = 256          # WGNN_STATUS_BYTES: status block at the start of every workspace
OPT_FUSED_FWD = 0           # WGNN_OPT_FUSED_FWD (wgnn_set_option): 0 never / 1 stash-less forwards / 2 every supported forward
OPT_GG_ROLE_SPLIT, OPT_GG_GEMM_PRIO, OPT_BWD2_CHUNKS, OPT_BIG_GEMM, OPT_GEMM32_FORM = 1, 2, 3, 4, 5   # measurement aids (include/windgnn.h): same results, another schedule


class Dims(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("S", C.c_int32), ("F", C.c_int32), ("H", C.c_int32),
                ("math", C.c_int32), ("adj_format", C.c_int32), ("nnz", C.c_int32), ("io", C.c_int32)]


_SLOTS = ("conv1_weight", "conv1_bias", "conv2_weight", "conv2_bias", "w_ih", "w_hh", "b_ih", "b_hh")


class Params(C.Structure):
    # wgnn_params: the 8 tensors + the optional caller-kept images of W_ih (NULL = rebuilt inside every call)
    _fields_ = [(n, C.c_void_p) for n in _SLOTS] + [("prepared", C.c_void_p)]


class Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _SLOTS]


class Adam(C.Structure):
    # wgnn_adam
    _fields_ = [("exp_avg", Grads), ("exp_avg_sq", Grads), ("step", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float)]


BWD_DEFER = 16              # WGNN_BWD_DEFER
FINISH_ADAM_GRU = 16        # WGNN_FINISH_ADAM_GRU / _CONV: wgnn_finish's optimiser step of one tensor family only
FINISH_ADAM_CONV = 32


EXPORTS = {
    "wgnn_version": (C.c_int, []),
    "wgnn_strerror": (C.c_char_p, [C.c_int]),
    "wgnn_set_option": (C.c_int, [C.c_int, C.c_int]),
    "wgnn_get_option": (C.c_int, [C.c_int]),
    "wgnn_workspace_bytes": (C.c_size_t, [C.POINTER(Dims)]),
    "wgnn_stash_bytes": (C.c_size_t, [C.POINTER(Dims)]),
    "wgnn_fwd": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "wgnn_fwd_loss": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wgnn_fwd_last": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_float, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wgnn_bwd": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                           C.c_void_p, C.POINTER(Grads), C.c_void_p, C.c_size_t, C.c_void_p]),
    "wgnn_bwd_part": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(Grads), C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "wgnn_bwd_mse_part": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_void_p, C.c_void_p, C.POINTER(Grads), C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_int]),
    "wgnn_finish": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Grads), C.c_int, C.POINTER(Adam), C.c_void_p,
                              C.c_size_t, C.c_void_p]),
    "wgnn_prepared_bytes": (C.c_size_t, [C.POINTER(Dims)]),
    "wgnn_prepare_weights": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p]),
    "wgnn_gcn_layer_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "wgnn_gcn_layer_fwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "wgnn_gcn_layer_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "wgnn_gru_fwd": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_size_t, C.c_void_p]),
    "wgnn_gru_bwd": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(Grads), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wgnn_gcn_layer_csr_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "wgnn_gcn_layer_csr_fwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "wgnn_gcn_layer_csr_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]),
    "wgnn_mse_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "wgnn_make_windows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wgnn_predict_last": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                    C.c_void_p]),
    "wgnn_profile_enable": (C.c_int, [C.c_int]),
    "wgnn_profile_read": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "wgnn_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library or raise.  Never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "windgnn_amd: %s is missing. Build it with `python -m windgnn_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.wgnn_version() < 122:
        raise RuntimeError("windgnn_amd: libwindgnn_hip.so is too old")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().wgnn_strerror(status).decode()
        raise RuntimeError("windgnn_amd: %s failed: %s (status %d)" % (what, msg, status))


def set_option(key: int, value: int) -> int:
    """wgnn_set_option: returns the previous value; raises on an unknown key / value."""
    prev = load().wgnn_set_option(key, value)
    if prev < 0:
        check(prev, "wgnn_set_option(%d, %d)" % (key, value))
    return prev


def get_option(key: int) -> int:
    v = load().wgnn_get_option(key)
    if v < 0:
        check(v, "wgnn_get_option(%d)" % key)
    return v


def profile_enable(on: bool) -> None:
    check(load().wgnn_profile_enable(1 if on else 0), "wgnn_profile_enable")


def profile_read():
    """[{name, ms, launches, flops, bytes}] accumulated since profile_enable(True)."""
    lib = load()
    out = []
    i = 0
    while True:
        name = C.create_string_buffer(128)
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        if lib.wgnn_profile_read(i, name, 128, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)) != 0:
            break
        out.append({"name": name.value.decode(), "ms": ms.value, "launches": n.value, "flops": fl.value,
                    "bytes": by.value})
        i += 1
    return out
