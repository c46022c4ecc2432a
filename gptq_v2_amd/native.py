"""The reference's native operator boundary (``package/gptq/__init__.py:6-28`` and
``gptq_triton/accumulate_hessian.py:134-149``) with the same names, argument order and in-place
behaviour, served by libgptq_amd.so."""
from __future__ import annotations

import torch

from . import ops


def accumulate_hessian(mat_hessian: torch.Tensor, mat_input: torch.Tensor) -> None:
    """H += X^T X, full product (the CUTLASS op's contract).  fp16 and bf16 inputs only; any
    other dtype raises NotImplementedError as the reference does (package/gptq/__init__.py:12-13)."""
    if mat_input.dtype not in (torch.float16, torch.bfloat16):
        raise NotImplementedError
    ops.xtx_accumulate(mat_hessian, mat_input, save_lower_only=False, compute_lower_only=False)


def accumulate_hessian_triton(mat_hessian: torch.Tensor, mat_input: torch.Tensor, save_lower_only: bool,
                              compute_lower_only: bool = True) -> None:
    """Signature of the reference's Triton op (gptq_triton/accumulate_hessian.py:134-149)."""
    assert compute_lower_only or not save_lower_only, \
        "compute_lower_only must be True when save_lower_only is True"
    ops.xtx_accumulate(mat_hessian, mat_input, save_lower_only=save_lower_only,
                       compute_lower_only=compute_lower_only)


def gptq_quantize_range(quant: torch.Tensor, scale: torch.Tensor, out_q: torch.Tensor, qzero: torch.Tensor,
                        maxq: float, hessian_inv: torch.Tensor, weights: torch.Tensor, error: torch.Tensor,
                        a: int, b: int) -> None:
    """package/gptq/__init__.py:16-28: (C, R) fp32 layouts, ``weights`` updated in place,
    ``out_q`` int16 or a meta tensor for "no integer output"."""
    ops.quantize_range(quant, scale, out_q, qzero, maxq, hessian_inv, weights, error, a, b)
