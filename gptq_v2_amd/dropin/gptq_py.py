"""Stands in for the reference's ``gptq_py.py``."""
from gptq_v2_amd.gptq_py import GPTQ, HessianHook, gptq_quant, gptq_quant_set  # noqa: F401
