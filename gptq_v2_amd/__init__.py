"""MI355X-native GPTQ calibration + int4 inference hot path (HIP kernels behind a C ABI).

Importing the package loads gptq_v2_amd/lib/libgptq_amd.so and fails if it is missing:
there is no CPU or PyTorch fallback for the kernels.
"""
from . import _lib

_lib.load()

from . import ops  # noqa: E402
from .gptq_py import GPTQ, HessianHook, gptq_quant, gptq_quant_set  # noqa: E402
from .native import accumulate_hessian, accumulate_hessian_triton, gptq_quantize_range  # noqa: E402
from .quant import (HipLinear, QuantLinear, Quantizer, collate_quantizers, construct_matrix, dequantize,  # noqa: E402
                    quantize, reconstruct_nn_linear)

__all__ = ["ops", "GPTQ", "HessianHook", "gptq_quant", "gptq_quant_set", "accumulate_hessian", "accumulate_hessian_triton",
           "gptq_quantize_range", "HipLinear", "QuantLinear", "Quantizer", "collate_quantizers", "construct_matrix",
           "dequantize", "quantize", "reconstruct_nn_linear"]
