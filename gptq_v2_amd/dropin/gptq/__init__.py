"""Stands in for the reference's CUDA extension package ``gptq`` (package/gptq/__init__.py)."""
from gptq_v2_amd.native import accumulate_hessian, gptq_quantize_range  # noqa: F401
