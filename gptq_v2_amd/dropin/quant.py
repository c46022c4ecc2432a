"""Stands in for the reference's ``quant.py``: every public name of it."""
from gptq_v2_amd.quant import (EPSILON, QuantLinear, Quantizer, collate_quantizers, construct_matrix,  # noqa: F401
                               construct_matrix_2, dequantize, dequantize2, dequantize2_quantized2,
                               dequantize_quantized, quantize, quantize2, reconstruct_nn_linear)
