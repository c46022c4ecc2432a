"""ctypes binding of libgptq_amd.so (the C ABI declared in include/gptq_amd.h).

There is no fallback: if the shared library is missing or a symbol is absent the import of
any op fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))


class GridOptions(ctypes.Structure):
    """struct gq_grid_options"""
    _fields_ = [("sym", c_int), ("scale_bits", c_int), ("mse", c_int), ("shrink", c_void_p), ("n_shrink", c_int),
                ("norm", c_float)]


# GQ_LIB_PATH: another build of the same library (the instrumented copies tools/ makes); same symbols required
LIB_PATH = os.environ.get("GQ_LIB_PATH") or os.path.join(_HERE, "lib", "libgptq_amd.so")

# name -> (restype, argtypes); must list every function include/gptq_amd.h declares
SIGNATURES = {
    "gq_last_error": (c_char_p, []),
    "gq_version": (c_int, []),
    "gq_xtx_accumulate": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_void_p]),
    "gq_xtx_accumulate_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "gq_set_invert_split": (c_int, [c_int]),
    "gq_device_info": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gq_hessian_invert_workspace": (c_size_t, [c_int64]),
    "gq_hessian_invert": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_float, c_int,
                                  c_int, c_void_p, c_size_t, ctypes.POINTER(c_int), c_void_p]),
    "gq_hessian_invert_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_float, c_int,
                                        c_void_p, c_size_t, c_void_p, c_void_p]),
    "gq_hessian_invert_finish": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_size_t,
                                         ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_int), c_void_p]),
    "gq_hessian_invert_static_max_c": (c_int64, []),
    "gq_hessian_invert_static": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_float, c_int,
                                         c_int, ctypes.POINTER(c_int), c_void_p]),
    "gq_find_params": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p,
                               c_void_p]),
    "gq_quantize_range": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "gq_quantize_layer_workspace": (c_size_t, [c_int64, c_int64]),
    "gq_quantize_layer": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64,
                                  ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), c_int, c_int,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "gq_quantize_rtn": (c_int, [c_void_p, c_int, c_int64, c_int64, ctypes.POINTER(ctypes.c_int32),
                                ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "gq_quantize_layer_ex": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64,
                                     ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), c_int,
                                     ctypes.POINTER(ctypes.c_int32), c_int, c_int, ctypes.POINTER(GridOptions),
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "gq_quant_metrics": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                 c_void_p, c_void_p]),
    "gq_dequantize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p,
                              c_int, c_void_p, c_void_p]),
    "gq_pack_rows": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "gq_unpack_rows": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "gq_permute_columns": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gq_pack_lower": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "gq_unpack_lower": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "gq_w4a16_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                              c_int, c_void_p]),
    "gq_w4a16_gemm_dt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                 c_int, c_int, c_void_p]),
    "gq_wq_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                           c_int, c_int, c_int, c_void_p]),
    "gq_wq_expand": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "gq_set_wq_kernel": (c_int, [c_int]),
    "gq_set_w4_tile": (c_int, [c_int]),
    "gq_set_quant_super": (c_int, [c_int]),
    "gq_xtx_split_hint": (c_int, [c_int64, c_int64, c_int]),
    "gq_xtx_accumulate_split": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_void_p]),
    "gq_xtx_fold": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "gq_gemm_nt_sub_lower": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "gq_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_float, c_void_p]),
    "gq_silu_mul": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "gq_rope_inplace": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "gq_xtx_tail_workspace": (c_size_t, [c_int64]),
    "gq_xtx_accumulate_tail": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "gq_xtx_fold_tail": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "gq_gemm_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
}

ABI_VERSION = 3   # GQ_ABI_VERSION of include/gptq_amd.h this table was written against

_lib = None


class GqError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is required (build it with "
                f"`python gptq_v2_amd/build.py`); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.gq_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} implements interface version {lib.gq_version()}, this binding expects "
                              f"{ABI_VERSION}: rebuild it (python gptq_v2_amd/build.py)")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gq_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise GqError(f"{what} failed (code {rc}): {msg}")
