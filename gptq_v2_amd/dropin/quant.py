"""Stands in for the reference's ``quant.py``."""
from gptq_v2_amd.quant import (EPSILON, QuantLinear, Quantizer, collate_quantizers, construct_matrix,  # noqa: F401
                               dequantize, dequantize_quantized, quantize, reconstruct_nn_linear)
