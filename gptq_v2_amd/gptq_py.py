"""Mirror of the reference's ``gptq_py.py`` (HessianHook, gptq_quant) on the HIP kernels.

Same call signatures, state fields, result dictionary and metrics as gptq_py.py:9-251.  Every
tensor stays on the GPU; the arithmetic is in libgptq_amd.so.  There is no CPU path: CPU tensors
are rejected.

``GPTQ`` at the bottom is a thin adapter with the classic ``add_batch`` / ``fasterquant`` names
over the same two objects.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .quant import _group_list


class HessianHook:
    """Running Hessian H = sum X^T X of one Linear's inputs and its inverse factor
    (gptq_py.py:9-66)."""

    def __init__(self) -> None:
        self.hessian: Optional[torch.Tensor] = None
        self.n_samples: int = 0
        self.hessian_inv: Optional[torch.Tensor] = None
        self.perm: Optional[torch.Tensor] = None
        self.perm_inv: Optional[torch.Tensor] = None
        self.tries: int = 0

    @torch.no_grad()
    def add_batch(self, inp: torch.Tensor, use_kernel: bool = True) -> None:
        """inp (..., N, D) fp16/bf16.  ``n_samples`` counts sequences, a 2-D input is one sequence
        (gptq_py.py:25-27).  ``use_kernel`` is accepted for signature compatibility: both of the
        reference's branches compute H += X^T X with fp32 accumulation, which is what the HIP
        kernel does; there is no second implementation to switch to."""
        assert self.hessian_inv is None
        assert not hasattr(inp, "fake_mode")
        if inp.dim() <= 2:
            inp = inp[None]
        self.n_samples += len(inp)
        d = inp.size(-1)
        if self.hessian is None:
            self.hessian = torch.zeros(d, d, dtype=torch.float32, device=inp.device)
        x = inp.flatten(end_dim=-2)
        if not x.is_contiguous():
            x = x.contiguous()
        ops.xtx_accumulate(self.hessian, x, save_lower_only=False, compute_lower_only=True)

    @torch.no_grad()
    def invert(self, damp_ratio: float = 1e-2, act_order: bool = True) -> torch.Tensor:
        """gptq_py.py:38-66.  Leaves ``hessian`` scaled, permuted and damped, sets ``perm`` /
        ``perm_inv`` (int64, as torch.argsort returns them) and returns ``hessian_inv``."""
        u, perm, perm_inv, tries = ops.hessian_invert(self.hessian, float(self.n_samples), damp_ratio, act_order,
                                                      max_try=100)
        self.hessian_inv = u
        self.tries = tries
        if act_order:
            self.perm = perm.to(torch.int64)
            self.perm_inv = perm_inv.to(torch.int64)
        return self.hessian_inv


@torch.no_grad()
def gptq_quant(
        weight: torch.Tensor,
        hessian_hook: Optional[HessianHook],
        group_sizes: torch.Tensor,
        group_bit_widths: torch.Tensor,
        scale_bit_width: torch.Tensor = None,
        gptq_use_kernel: bool = True,
        gptq_block_sizes: torch.Tensor = None,
        quant_symmetric: bool = False,
        quant_mse: bool = True,
        quant_max_shrink: float = .8,
        quant_n_grid: int = 100,
        quant_norm: float = 2.4,
        quant_use_kernel: bool = False,
        save_device: torch.device = torch.device('cpu'),
) -> dict:
    """gptq_py.py:69-251.  ``gptq_use_kernel=True`` selects the semantics of the reference's
    native loop (grid from the unmodified weights, fp16 dequantisation), ``False`` those of its
    pure-PyTorch loop (grid from the error-compensated weights, fp32 dequantisation);
    ``hessian_hook=None`` rounds to nearest.  Returns {'quant_meta': ..., 'metrics': ...}."""
    if scale_bit_width is not None:
        raise NotImplementedError("EXL2 scale quantisation (scale_bit_width) is not supported")
    if quant_mse:
        raise NotImplementedError("quant_mse grid search is not supported (llama.py runs with quant_mse=False)")
    n_rows, n_columns = weight.shape
    dtype = weight.dtype
    device = weight.device
    gs = _group_list(group_sizes)
    gb = _group_list(group_bit_widths)
    if hessian_hook is not None and not gptq_use_kernel:
        if gptq_block_sizes is None or _group_list(gptq_block_sizes) != gs:
            raise NotImplementedError("the PyTorch-loop semantics are implemented for gptq_block_sizes == group_sizes")
    w = weight if weight.is_contiguous() else weight.contiguous()

    perm32 = None
    u = None
    if hessian_hook is not None:
        u = hessian_hook.hessian_inv
        if hessian_hook.perm is not None:
            perm32 = hessian_hook.perm.to(torch.int32).contiguous()
        # gptq_py.py:95-96 zeroes columns whose Hessian diagonal is 0; after invert() no diagonal
        # entry is 0 any more (dead ones were set to 1, then damped), so nothing is zeroed there either.
    res = ops.quantize_layer(w, perm32, u, gs, gb, sym=quant_symmetric,
                             semantics=ops.SEM_KERNEL if gptq_use_kernel else ops.SEM_PYTHON,
                             want_quant=True, want_error=hessian_hook is not None)
    sums = ops.quant_metrics(w, perm32, u, res["quant_t"], res["error_t"]).tolist()
    n = float(n_rows * n_columns)
    gptq_error = sums[0] / n
    metrics = {
        "gptq_error": gptq_error,
        "gptq_norm": sums[1] / n if hessian_hook is not None else gptq_error,
        "l2.0_error": sums[2] / n,
        # the reference takes this mean in the weight's 16-bit dtype (gptq_py.py:237)
        "l2.0_norm": torch.tensor(sums[3] / n, dtype=torch.float64).to(dtype).item(),
    }

    if perm32 is not None:
        perm = hessian_hook.perm.to(torch.int16)
        perm_inv = hessian_hook.perm_inv.to(torch.int16)
    else:
        perm_inv = torch.arange(n_columns, dtype=torch.int16, device=device)
    quant_meta = {
        "qweight": res["qweight"].to(device=save_device),
        "scale": res["scale"].to(device=save_device),
        "qzero": res["qzero"].to(device=save_device),
        "qscale": None,
        "sscale": None,
        "perm_inv": perm_inv.to(device=save_device),
        "group_sizes": torch.as_tensor(gs, dtype=torch.int16, device=save_device),
        "group_bit_widths": torch.as_tensor(gb, dtype=torch.uint8, device=save_device),
        "scale_bit_width": None,
    }
    return {"quant_meta": quant_meta, "metrics": metrics}


class GPTQ:
    """``add_batch`` / ``fasterquant`` names over HessianHook + gptq_quant for one Linear."""

    def __init__(self, layer: torch.nn.Linear) -> None:
        self.layer = layer
        self.hook = HessianHook()

    def add_batch(self, inp: torch.Tensor, out: torch.Tensor = None) -> None:
        self.hook.add_batch(inp)

    def fasterquant(self, blocksize: int = 128, percdamp: float = .01, groupsize: int = 128, actorder: bool = True,
                    bits: int = 4, sym: bool = False, static_groups: bool = True,
                    save_device: torch.device = None) -> dict:
        w = self.layer.weight.data
        c = w.shape[1]
        gsz = c if groupsize in (-1, None) else groupsize
        if c % gsz:
            raise ValueError("in_features must be a multiple of the group size")
        group_sizes = torch.full([c // gsz], gsz, dtype=torch.int32)
        group_bits = torch.full([c // gsz], bits, dtype=torch.int32)
        self.hook.invert(damp_ratio=percdamp, act_order=actorder)
        return gptq_quant(w, self.hook, group_sizes, group_bits, gptq_use_kernel=static_groups,
                          gptq_block_sizes=group_sizes, quant_symmetric=sym, quant_mse=False,
                          save_device=save_device if save_device is not None else w.device)
