"""Stands in for the reference's ``gptq_triton/accumulate_hessian.py`` (no Triton involved)."""
from gptq_v2_amd.native import accumulate_hessian_triton as accumulate_hessian  # noqa: F401
