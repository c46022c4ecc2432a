"""Mirror of the reference's ``quant.py`` surface on top of the HIP kernels.

Same names, argument meaning and result types as the reference (quant.py:10-30, 72-118,
167-183, 186-237, 262-279), plus ``QuantLinear`` (pack / forward), which the reference does not
have: it reconstructs a dense fp16 ``nn.Linear`` (llama.py:172-177) where this module can also run
the packed int4 x fp16 GEMM.

Not mirrored: the EXL2 quadratic scale quantisation (quantize2 / dequantize2 / construct_matrix_2,
quant.py:40-69, 240-259) and ``Quantizer.mse`` (quant.py:120-165); the reference's driver switches
both off (llama.py:153, 157) and asking for them here raises NotImplementedError.

The element-wise helpers below work on tensors of any device, as the reference's do; the hot
paths (gptq_quant's loops, reconstruct_nn_linear, QuantLinear) go through libgptq_amd.so and
require a GPU.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops

EPSILON = 1e-12


def quantize(x: torch.Tensor, scale: torch.Tensor, qzero: torch.Tensor, maxq) -> torch.Tensor:
    """clamp(round(x / scale) + qzero, 0, maxq), round-half-even (quant.py:10-19)."""
    return torch.clamp(torch.round(x / scale) + qzero, 0.0, maxq)


def dequantize(qx: torch.Tensor, scale: torch.Tensor, qzero: torch.Tensor) -> torch.Tensor:
    """(qx - qzero) * scale (quant.py:22-30)."""
    return (qx - qzero) * scale


def dequantize_quantized(x, scale, qzero, maxq):
    return dequantize(quantize(x, scale, qzero, maxq), scale, qzero)


class Quantizer:
    """Per-row quantisation grid over the last dimension (quant.py:72-183)."""

    def __init__(self, scale: torch.Tensor = None, qzero: torch.Tensor = None, maxq=None,
                 qscale=None, sscale=None, smaxq=None):
        if qscale is not None or sscale is not None or smaxq is not None:
            raise NotImplementedError("EXL2 scale quantisation (qscale/sscale) is not supported")
        self.scale = scale
        self.qzero = qzero
        self.maxq = maxq
        self.qscale = None
        self.sscale = None
        self.smaxq = None

    def find_params(self, x: torch.Tensor, bit_width, sym: bool = False, scale_bit_width=None) -> None:
        if scale_bit_width is not None and scale_bit_width > 0:
            raise NotImplementedError("EXL2 scale quantisation (scale_bit_width) is not supported")
        self.maxq = 2.0 ** bit_width - 1.0
        if sym:
            self.scale = x.abs().amax(dim=-1, keepdim=True) * (2.0 / self.maxq) + EPSILON
            mid = torch.round(torch.as_tensor((self.maxq + 1.0) * 0.5))
            self.qzero = torch.full_like(self.scale, float(mid))
        else:
            hi = torch.relu(x.amax(dim=-1, keepdim=True))
            lo = -torch.relu(-x.amin(dim=-1, keepdim=True))
            self.scale = (hi - lo) / self.maxq + EPSILON
            self.qzero = torch.round(-lo / self.scale)

    def mse(self, *args, **kwargs) -> None:
        raise NotImplementedError("Quantizer.mse grid search is not supported (llama.py runs with quant_mse=False)")

    def quantize(self, x):
        return quantize(x, self.scale, self.qzero, self.maxq)

    def dequantize(self, qx):
        return dequantize(qx, self.scale, self.qzero)

    def dequantize_quantized(self, x):
        return dequantize_quantized(x, self.scale, self.qzero, self.maxq)


def collate_quantizers(*args) -> dict:
    """Concatenate the tensor attributes of several Quantizers along the last dimension;
    0-dim attributes are stacked (quant.py:186-219)."""
    buckets: dict = {}
    for q in args:
        for name in dir(q):
            if name.startswith("_"):
                continue
            value = getattr(q, name)
            if isinstance(value, torch.Tensor):
                buckets.setdefault(name, []).append(value)
    return {k: (torch.cat(v, dim=-1) if v[0].dim() else torch.stack(v)) for k, v in buckets.items()}


def _group_list(group_sizes) -> list:
    return [int(v) for v in (group_sizes.tolist() if isinstance(group_sizes, torch.Tensor) else group_sizes)]


def construct_matrix(qweight: torch.Tensor, scale: torch.Tensor, qzero: torch.Tensor,
                     group_sizes: torch.Tensor) -> torch.Tensor:
    """(R, C) weight in ``scale``'s dtype from per-group integer levels (quant.py:222-237);
    columns stay in the order of ``qweight``."""
    gs = _group_list(group_sizes)
    dev = qweight.device
    qz = qzero.expand(scale.shape)
    return ops.dequantize(qweight.to(torch.uint8).contiguous(), scale.contiguous(),
                          qz.to(device=dev, dtype=torch.uint8).contiguous(), None, gs)


def reconstruct_weight(quant_meta: dict, dtype: torch.dtype, device) -> torch.Tensor:
    qweight = quant_meta["qweight"].to(device=device, dtype=torch.uint8).contiguous()
    scale = quant_meta["scale"].to(device=device, dtype=dtype).contiguous()
    qzero = quant_meta["qzero"].to(device=device, dtype=torch.uint8).contiguous()
    gs = _group_list(quant_meta["group_sizes"])
    perm_inv = quant_meta.get("perm_inv")
    if perm_inv is not None:
        perm_inv = perm_inv.to(device=device, dtype=torch.int32).contiguous()
    return ops.dequantize(qweight, scale, qzero, perm_inv, gs)


def reconstruct_nn_linear(quant_meta: dict, dtype: torch.dtype, device: torch.device = None) -> nn.Linear:
    """Dense ``nn.Linear`` (no bias) from a ``quant_meta`` dict (quant.py:262-279).  The
    dequantisation runs on the GPU, so ``device`` must be a GPU (the reference defaults to CPU)."""
    if quant_meta.get("sscale") is not None:
        raise NotImplementedError("EXL2 quant_meta (sscale/qscale) is not supported")
    if device is None:
        device = quant_meta["qweight"].device
    weight = reconstruct_weight(quant_meta, dtype, device)
    with torch.device("meta"):
        lin = nn.Linear(weight.shape[1], weight.shape[0], bias=False, dtype=dtype)
    lin.weight = nn.Parameter(weight, requires_grad=False)
    lin.eval()
    return lin


class QuantLinear(nn.Module):
    """Packed int4 Linear: ``pack`` takes a ``quant_meta`` dict, ``forward`` runs the
    W4A16 dequant-GEMM (activations gathered into act-order first when the layer has one).

    Packed state, laid out so that a GEMM tile reads contiguous memory: ``qweight`` int32
    (C/64, R, 8) (pair-interleaved nibbles, k-block major), ``scales`` fp16 (G, R), ``qzeros`` uint8
    (G, R), ``perm`` int32 (C) or None.
    ``unpack`` returns the original one-value-per-byte ``quant_meta`` tensors bit for bit.
    """

    def __init__(self, in_features: int, out_features: int, group_size: int = 128, bits: int = 4):
        super().__init__()
        if bits != 4:
            raise NotImplementedError("QuantLinear.forward is implemented for 4-bit weights")
        self.in_features, self.out_features = in_features, out_features
        self.group_size, self.bits = group_size, bits
        self.register_buffer("qweight", None)
        self.register_buffer("scales", None)
        self.register_buffer("qzeros", None)
        self.register_buffer("perm", None)

    @classmethod
    def from_quant_meta(cls, quant_meta: dict, device) -> "QuantLinear":
        r, c = quant_meta["qweight"].shape
        gs = _group_list(quant_meta["group_sizes"])
        bits = set(_group_list(quant_meta["group_bit_widths"]))
        if len(set(gs)) != 1 or len(bits) != 1:
            raise NotImplementedError("QuantLinear needs one group size and one bit width")
        layer = cls(c, r, gs[0], bits.pop())
        layer.pack(quant_meta, device)
        return layer

    @torch.no_grad()
    def pack(self, quant_meta: dict, device=None) -> None:
        device = device if device is not None else quant_meta["qweight"].device
        q = quant_meta["qweight"].to(device=device, dtype=torch.uint8).contiguous()
        if q.shape != (self.out_features, self.in_features):
            raise ValueError("quant_meta does not match this layer's shape")
        if self.in_features % 64:
            raise NotImplementedError("QuantLinear needs in_features to be a multiple of 64")
        self.qweight = ops.pack_rows(q, self.bits, ops.PACK_W4_KBLOCK)
        self.scales = quant_meta["scale"].to(device=device, dtype=torch.float16).t().contiguous()
        self.qzeros = quant_meta["qzero"].to(device=device, dtype=torch.uint8).t().contiguous()
        perm_inv = quant_meta.get("perm_inv")
        if perm_inv is None:
            self.perm = None
        else:
            pi = perm_inv.to(device=device, dtype=torch.int64)
            ident = torch.arange(pi.numel(), device=device)
            # perm = argsort(perm_inv): scatter, exact because perm_inv is a permutation
            perm = torch.empty_like(pi)
            perm[pi] = ident
            self.perm = None if torch.equal(pi, ident) else perm.to(torch.int32).contiguous()

    @torch.no_grad()
    def unpack(self) -> dict:
        q = ops.unpack_rows(self.qweight, self.bits, self.in_features, ops.PACK_W4_KBLOCK)
        c = self.in_features
        if self.perm is None:
            perm_inv = torch.arange(c, dtype=torch.int16, device=q.device)
        else:
            perm_inv = torch.empty(c, dtype=torch.int64, device=q.device)
            perm_inv[self.perm.long()] = torch.arange(c, device=q.device)
            perm_inv = perm_inv.to(torch.int16)
        g = c // self.group_size
        return {"qweight": q, "scale": self.scales.t().contiguous(), "qzero": self.qzeros.t().contiguous(),
                "qscale": None, "sscale": None,
                "perm_inv": perm_inv,
                "group_sizes": torch.full([g], self.group_size, dtype=torch.int16, device=q.device),
                "group_bit_widths": torch.full([g], self.bits, dtype=torch.uint8, device=q.device),
                "scale_bit_width": None}

    @torch.no_grad()
    def permute_input(self, x: torch.Tensor) -> torch.Tensor:
        """x[..., perm]: the activation gather of act-order.  Layers quantised against the same
        Hessian (q/k/v, gate/up) share ``perm``, so their caller can do this once and use
        ``forward_permuted`` on each of them."""
        if x.dtype != torch.float16:
            raise NotImplementedError("QuantLinear takes float16 activations")
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        if self.perm is not None:
            x2 = ops.permute_columns(x2, self.perm)
        return x2.reshape(shape)

    @torch.no_grad()
    def forward_permuted(self, x_permuted: torch.Tensor) -> torch.Tensor:
        shape = x_permuted.shape
        x2 = x_permuted.reshape(-1, shape[-1])
        if x2.dtype != torch.float16:
            raise NotImplementedError("QuantLinear takes float16 activations")
        y = ops.w4a16_gemm(x2.contiguous(), self.qweight, self.scales, self.qzeros, self.group_size)
        return y.reshape(*shape[:-1], self.out_features)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_permuted(self.permute_input(x))
