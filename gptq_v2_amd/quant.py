"""Mirror of the reference's ``quant.py`` surface on top of the HIP kernels.

Same names, argument meaning and result types as the reference (quant.py:10-30, 72-118,
167-183, 186-237, 262-279), plus ``QuantLinear`` (pack / forward), which the reference does not
have: it reconstructs a dense fp16 ``nn.Linear`` (llama.py:172-177) where this module can also run
the packed int4 x fp16 GEMM.

The EXL2 quadratic scale quantisation (quantize2 / dequantize2 / construct_matrix_2, quant.py:40-69,
240-259) and ``Quantizer.mse`` (quant.py:120-165, its PyTorch branch; the other one calls an
extension the reference itself does not import, quant.py:4) are mirrored too; the reference's driver
switches both off (llama.py:153, 157).

The element-wise helpers and ``Quantizer`` below are plain tensor arithmetic on whatever device the
tensors live on, as the reference's are; the hot paths (gptq_quant's loops with their grids,
reconstruct_nn_linear, QuantLinear) go through libgptq_amd.so and require a GPU.
"""
from __future__ import annotations

from typing import Optional

import os

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


def quantize2(x: torch.Tensor, scale: torch.Tensor, maxq) -> torch.Tensor:
    """Quadratic quantisation of positive numbers (a column of scales, EXL2): the level q in
    1..maxq+1 with q^2 * scale nearest in the square-root domain (quant.py:40-51)."""
    return torch.clamp(torch.round(torch.sqrt(x / scale)), 1.0, maxq + 1.0)


def dequantize2(qx: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """qx^2 * scale (quant.py:54-62)."""
    return qx * qx * scale


def dequantize2_quantized2(x, scale, maxq):
    return dequantize2(quantize2(x, scale, maxq), scale)


class Quantizer:
    """Per-row quantisation grid over the last dimension (quant.py:72-183); with
    ``scale_bit_width`` the scales themselves sit on a quadratic grid ``qscale^2 * sscale``."""

    def __init__(self, scale: torch.Tensor = None, qzero: torch.Tensor = None, maxq=None,
                 qscale=None, sscale=None, smaxq=None):
        self.scale = scale      # (..., R, 1)
        self.qzero = qzero      # (..., R, 1)
        self.maxq = maxq        # ()
        self.qscale = qscale    # (..., R, 1)
        self.sscale = sscale    # (..., 1, 1)
        self.smaxq = smaxq      # ()
        # quant.py:84-87: whichever of scale / qscale is missing follows from the other
        if scale is None and qscale is not None and sscale is not None:
            self.scale = dequantize2(qscale, sscale)
        elif qscale is None and scale is not None and sscale is not None and smaxq is not None:
            self.qscale = quantize2(scale, sscale, smaxq)

    def _requantize_scale(self) -> None:
        self.qscale = quantize2(self.scale, self.sscale, self.smaxq)
        self.scale = dequantize2(self.qscale, self.sscale)

    def _sscale_from_scale(self) -> torch.Tensor:
        return self.scale.amax(dim=-2, keepdim=True) / (self.smaxq + 1.0) ** 2.0 + EPSILON

    def find_params(self, x: torch.Tensor, bit_width, sym: bool = False, scale_bit_width=None) -> None:
        """quant.py:89-118.  bit_width / scale_bit_width: 0-dim tensors (or numbers)."""
        exl2 = scale_bit_width is not None and scale_bit_width > 0
        self.maxq = 2.0 ** bit_width - 1.0
        if sym or exl2:
            self.scale = x.abs().amax(dim=-1, keepdim=True) * (2.0 / self.maxq) + EPSILON
            mid = torch.round(torch.as_tensor((self.maxq + 1.0) * 0.5))
            self.qzero = torch.full_like(self.scale, float(mid))
            if exl2:
                self.smaxq = 2.0 ** scale_bit_width - 1.0
                self.sscale = self._sscale_from_scale()
                self._requantize_scale()
        else:
            hi = torch.relu(x.amax(dim=-1, keepdim=True))
            lo = -torch.relu(-x.amin(dim=-1, keepdim=True))
            self.scale = (hi - lo) / self.maxq + EPSILON
            self.qzero = torch.round(-lo / self.scale)

    def mse(self, x: torch.Tensor, max_shrink: float = .8, n_grid: int = 100, norm: float = 2.4,
            use_kernel: bool = False) -> None:
        """Shrinks every row's scale to the candidate with the smallest |error|^norm summed over the
        row, then (EXL2) does the same for the step of the scale grid (quant.py:120-165)."""
        if use_kernel:
            raise NotImplementedError("Quantizer.mse(use_kernel=True) calls exllamav2_ext, which the reference does "
                                      "not import either (quant.py:4)")
        p = 1.0 - torch.arange(0.0, max_shrink, 1.0 / n_grid, device=x.device)   # (Q)
        xq = x[..., None]
        cand = dequantize_quantized(xq, self.scale[..., None] * p, self.qzero[..., None], self.maxq)   # (..., R, C, Q)
        best = (cand - xq).abs().pow(norm).sum(dim=-2).argmin(dim=-1, keepdim=True)
        self.scale = self.scale * p[best]
        if self.sscale is not None:
            self.sscale = self._sscale_from_scale()
            sq = self.scale[..., None]
            cand = dequantize2_quantized2(sq, self.sscale[..., None] * p, self.smaxq)   # (..., R, 1, Q)
            best = (cand - sq).abs().pow(norm).sum(dim=-3).argmin(dim=-1, keepdim=True)
            self.sscale = self.sscale * p[best]
            self._requantize_scale()

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
    """Group sizes / bit widths as Python ints.  A tensor that ``with_host_list`` has tagged brings its values
    along, so a result dictionary kept on the GPU can be consumed without a device-to-host round trip."""
    if isinstance(group_sizes, torch.Tensor):
        host = getattr(group_sizes, "_host_list", None)
        return list(host) if host is not None else [int(v) for v in group_sizes.tolist()]
    return [int(v) for v in group_sizes]


def with_host_list(t: torch.Tensor, values) -> torch.Tensor:
    """Tags a small constant tensor with the Python values it was built from (see ``_group_list``)."""
    t._host_list = tuple(int(v) for v in values)
    return t


def construct_matrix(qweight: torch.Tensor, scale: torch.Tensor, qzero: torch.Tensor,
                     group_sizes: torch.Tensor) -> torch.Tensor:
    """(R, C) weight in ``scale``'s dtype from per-group integer levels (quant.py:222-237);
    columns stay in the order of ``qweight``."""
    gs = _group_list(group_sizes)
    dev = qweight.device
    qz = qzero.expand(scale.shape)
    return ops.dequantize(qweight.to(torch.uint8).contiguous(), scale.contiguous(),
                          qz.to(device=dev, dtype=torch.uint8).contiguous(), None, gs)


def exl2_scale(qscale: torch.Tensor, sscale: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(R, G) scales of an EXL2 ``quant_meta`` as the reference evaluates them when it rebuilds the
    weight (quant.py:84-85 with the operand types of quant.py:267-268): the squared level as an
    int16, times the group's step in ``dtype``."""
    q = qscale.to(torch.int16)
    return (q * q) * sscale.to(dtype).reshape(1, -1)


def construct_matrix_2(qweight: torch.Tensor, qzero: torch.Tensor, qscale: torch.Tensor, sscale: torch.Tensor,
                       group_sizes: torch.Tensor) -> torch.Tensor:
    """construct_matrix for EXL2 metadata: scale = qscale^2 * sscale (quant.py:240-259)."""
    return construct_matrix(qweight, exl2_scale(qscale, sscale, sscale.dtype), qzero, group_sizes)


def meta_scale(quant_meta: dict, dtype: torch.dtype, device) -> torch.Tensor:
    """The (R, G) scales reconstruct_nn_linear dequantises with (quant.py:266-272)."""
    if quant_meta.get("sscale") is not None:
        return exl2_scale(quant_meta["qscale"].to(device), quant_meta["sscale"].to(device), dtype).contiguous()
    return quant_meta["scale"].to(device=device, dtype=dtype).contiguous()


def reconstruct_weight(quant_meta: dict, dtype: torch.dtype, device) -> torch.Tensor:
    qweight = quant_meta["qweight"].to(device=device, dtype=torch.uint8).contiguous()
    scale = meta_scale(quant_meta, dtype, device)
    qzero = quant_meta["qzero"].to(device=device, dtype=torch.uint8).contiguous()
    gs = _group_list(quant_meta["group_sizes"])
    perm_inv = quant_meta.get("perm_inv")
    if perm_inv is not None:
        perm_inv = perm_inv.to(device=device, dtype=torch.int32).contiguous()
    return ops.dequantize(qweight, scale, qzero, perm_inv, gs)


class HipLinear(nn.Linear):
    """An ``nn.Linear`` without bias whose forward runs the hand-written dense MFMA kernel
    (``gq_gemm_nt``) when it can: 16-bit weight on a GPU, in_features a multiple of 64.  Anything
    else takes ``nn.Linear``'s own forward.  Same parameters and state dict as ``nn.Linear``; it is
    the module the reference's replay ``lin(h)`` (llama.py:176-177) runs through here."""

    @classmethod
    def from_weight(cls, weight: torch.Tensor) -> "HipLinear":
        with torch.device("meta"):
            lin = cls(weight.shape[1], weight.shape[0], bias=False, dtype=weight.dtype)
        lin.weight = nn.Parameter(weight, requires_grad=False)
        lin.eval()
        return lin

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w = self.weight
        if (self.bias is not None or w.device.type != "cuda" or w.dtype not in (torch.float16, torch.bfloat16)
                or x.dtype != w.dtype or x.device != w.device or self.in_features % 64 or torch.is_grad_enabled() and
                (x.requires_grad or w.requires_grad)):
            return super().forward(x)
        x2 = x.reshape(-1, self.in_features)
        y = ops.gemm_nt(x2 if x2.is_contiguous() else x2.contiguous(), w.data if w.is_contiguous() else w.data.contiguous())
        return y.reshape(*x.shape[:-1], self.out_features)


def reconstruct_nn_linear(quant_meta: dict, dtype: torch.dtype, device: torch.device = torch.device("cpu")) -> nn.Linear:
    """Dense ``nn.Linear`` (no bias) from a ``quant_meta`` dict (quant.py:262-279), placed on
    ``device`` (default CPU, as in the reference).  The dequantisation itself always runs on a GPU:
    on ``device`` when that is one, else on the GPU that holds ``qweight`` or the current GPU, and
    the finished weight is moved."""
    device = torch.device(device)
    if device.type == "cuda":
        compute = device
    elif quant_meta["qweight"].device.type == "cuda":
        compute = quant_meta["qweight"].device
    else:
        compute = torch.device("cuda", torch.cuda.current_device())
    weight = reconstruct_weight(quant_meta, dtype, compute).to(device)
    return HipLinear.from_weight(weight)   # an nn.Linear; on a GPU its forward is the MFMA kernel of this library


class QuantLinear(nn.Module):
    """Packed int4 Linear: ``pack`` takes a ``quant_meta`` dict, ``forward`` runs the
    W4A16 dequant-GEMM (activations gathered into act-order first when the layer has one).

    Everything ``reconstruct_nn_linear`` accepts with levels of at most 4 bits can be packed
    (quant.py:222-279): uniform or mixed group sizes and bit widths, plain or EXL2 scales
    (``qscale^2 * sscale``, evaluated exactly as the dense reconstruction evaluates them).  Mixed
    groups are re-expressed on the finest common grid -- groups of gcd(group sizes) columns, each
    carrying the scale and zero point of the group it lies in -- so the GEMM sees a uniform group
    size; that gcd must be a multiple of 32.  2- and 3-bit levels sit in the same 4-bit containers
    (the arithmetic (q - z) * scale is the same for every width up to 4) unless ``dense3`` packing
    is chosen for a layer whose levels all fit 3 bits.

    Packed state, laid out so that a GEMM tile reads contiguous memory: ``qweight`` int32
    (C/64, R, 8) (pair-interleaved nibbles, k-block major; (C/64, R, 6) for ``dense3``),
    ``scales`` fp16 or bf16 (G', R), ``qzeros`` uint8 (G', R) with G' = C / group_size fine groups,
    ``perm`` int32 (C) or None.  ``unpack`` returns the original one-value-per-byte ``quant_meta``
    tensors bit for bit.
    """

    def __init__(self, in_features: int, out_features: int, group_size: int = 128, bits: int = 4,
                 dtype: torch.dtype = torch.float16, group_sizes=None, group_bit_widths=None, dense3: bool = False):
        super().__init__()
        if dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError("QuantLinear works in float16 or bfloat16")
        self.dtype = dtype
        if bits not in (2, 3, 4):
            raise NotImplementedError("QuantLinear.forward is implemented for weights of 2 to 4 bits")
        if dense3 and bits > 3:
            raise ValueError("dense 3-bit storage needs levels of at most 3 bits")
        self.in_features, self.out_features = in_features, out_features
        self.group_size, self.bits = group_size, bits   # group_size: the GEMM's (fine) group; bits: the widest level
        self.dense3 = bool(dense3)
        # the groups as quant_meta describes them (None: in_features / group_size groups of `bits` bits)
        self.meta_group_sizes = None if group_sizes is None else [int(v) for v in group_sizes]
        self.meta_group_bits = None if group_bit_widths is None else [int(v) for v in group_bit_widths]
        self.register_buffer("qweight", None)
        self.register_buffer("scales", None)
        self.register_buffer("qzeros", None)
        self.register_buffer("perm", None)
        self.register_buffer("qscale", None)   # EXL2 metadata, kept only so that unpack() can hand it back
        self.register_buffer("sscale", None)
        self.register_buffer("scale_meta", None)   # quant_meta["scale"] of an EXL2 layer (not what the GEMM multiplies by)

    @classmethod
    def from_quant_meta(cls, quant_meta: dict, device, validate: bool = True, has_perm: Optional[bool] = None,
                        dense3: bool = False) -> "QuantLinear":
        r, c = quant_meta["qweight"].shape
        gs = _group_list(quant_meta["group_sizes"])
        gb = _group_list(quant_meta["group_bit_widths"])
        if max(gb) > 4:
            raise NotImplementedError("QuantLinear holds levels of at most 4 bits")
        import math
        fine = 0
        for v in gs:
            fine = math.gcd(fine, v)
        if fine % 32:
            raise NotImplementedError("QuantLinear needs group sizes with a common divisor that is a multiple of 32")
        sdt = quant_meta["scale"].dtype   # the model's 16-bit type (gptq_py.py:218)
        uniform = len(set(gs)) == 1 and len(set(gb)) == 1
        layer = cls(c, r, fine, max(2, max(gb)), sdt if sdt in (torch.float16, torch.bfloat16) else torch.float16,
                    None if uniform else gs, None if uniform else gb, dense3=dense3)
        layer.pack(quant_meta, device, validate=validate, has_perm=has_perm)
        return layer

    def _fine_index(self, device) -> Optional[torch.Tensor]:
        """For every fine group the index of the quant_meta group it lies in (None: they coincide)."""
        if self.meta_group_sizes is None:
            return None
        reps = torch.tensor([v // self.group_size for v in self.meta_group_sizes], device=device)
        return torch.repeat_interleave(torch.arange(len(self.meta_group_sizes), device=device), reps)

    @torch.no_grad()
    def pack(self, quant_meta: dict, device=None, validate: bool = True, has_perm: Optional[bool] = None) -> None:
        """``validate``: check that every level and zero point fits the bit width (one device-to-host
        round trip; a caller that packs what ``gptq_quant`` has just produced can skip it).
        ``has_perm``: whether ``perm_inv`` is a real permutation (True), the identity (False), or to be
        looked at (None: one more round trip)."""
        device = device if device is not None else quant_meta["qweight"].device
        q = quant_meta["qweight"].to(device=device, dtype=torch.uint8).contiguous()
        if q.shape != (self.out_features, self.in_features):
            raise ValueError("quant_meta does not match this layer's shape")
        if self.in_features % 64:
            raise NotImplementedError("QuantLinear needs in_features to be a multiple of 64")
        qz = quant_meta["qzero"].to(device=device, dtype=torch.uint8)
        if validate:
            worst = torch.maximum(q.max(), qz.max())
            if int(worst) >= (1 << self.bits):
                raise ValueError(f"quant_meta holds values outside {self.bits} bits")
        if self.dense3:
            self.qweight = ops.pack_rows(q, 3, ops.PACK_W3_KBLOCK)
        else:
            self.qweight = ops.pack_rows(q, 4, ops.PACK_W4_KBLOCK)
        scale = meta_scale(quant_meta, self.dtype, device)
        idx = self._fine_index(device)
        if idx is not None:
            if scale.shape[1] != len(self.meta_group_sizes):
                raise ValueError("quant_meta's scale does not match its group_sizes")
            scale, qz = scale[:, idx], qz.expand(scale.shape)[:, idx]
        self.scales = scale.t().contiguous()
        self.qzeros = qz.expand(self.out_features, self.scales.shape[0]).t().contiguous()
        if quant_meta.get("sscale") is not None:
            self.qscale = quant_meta["qscale"].to(device)
            self.sscale = quant_meta["sscale"].to(device)
            self.scale_meta = quant_meta["scale"].to(device)
        perm_inv = quant_meta.get("perm_inv")
        if perm_inv is None or has_perm is False:
            self.perm = None
        else:
            pi = perm_inv.to(device=device, dtype=torch.int64)
            ident = torch.arange(pi.numel(), device=device)
            # perm = argsort(perm_inv): scatter, exact because perm_inv is a permutation
            perm = torch.empty_like(pi)
            perm[pi] = ident
            if has_perm is None and torch.equal(pi, ident):
                self.perm = None
            else:
                self.perm = perm.to(torch.int32).contiguous()

    @torch.no_grad()
    def unpack(self) -> dict:
        if self.dense3:
            q = ops.unpack_rows(self.qweight, 3, self.in_features, ops.PACK_W3_KBLOCK)
        else:
            q = ops.unpack_rows(self.qweight, 4, self.in_features, ops.PACK_W4_KBLOCK)
        c = self.in_features
        if self.perm is None:
            perm_inv = torch.arange(c, dtype=torch.int16, device=q.device)
        else:
            perm_inv = torch.empty(c, dtype=torch.int64, device=q.device)
            perm_inv[self.perm.long()] = torch.arange(c, device=q.device)
            perm_inv = perm_inv.to(torch.int16)
        scale, qzero = self.scales.t(), self.qzeros.t()
        if self.meta_group_sizes is None:
            gs = [self.group_size] * (c // self.group_size)
            gb = [self.bits] * len(gs)
        else:
            gs, gb = self.meta_group_sizes, self.meta_group_bits
            first, at = [], 0
            for v in gs:   # the first fine group of every quant_meta group
                first.append(at)
                at += v // self.group_size
            first = torch.tensor(first, device=q.device)
            scale, qzero = scale[:, first], qzero[:, first]
        if self.scale_meta is not None:
            scale = self.scale_meta
        return {"qweight": q, "scale": scale.contiguous(), "qzero": qzero.contiguous(),
                "qscale": self.qscale, "sscale": self.sscale,
                "perm_inv": perm_inv,
                "group_sizes": torch.tensor(gs, dtype=torch.int16, device=q.device),
                "group_bit_widths": torch.tensor(gb, dtype=torch.uint8, device=q.device),
                "scale_bit_width": None}

    def packed_state(self) -> dict:
        """The layer as tensors, numbers, lists and None only (a file of these loads with
        ``torch.load(weights_only=True)``): the packed words as they lie in memory -- 4 or 3 bits per
        weight -- instead of the one byte per level of ``quant_meta``."""
        return {"in_features": self.in_features, "out_features": self.out_features, "group_size": self.group_size,
                "bits": self.bits, "dense3": self.dense3, "dtype": "float16" if self.dtype == torch.float16 else "bfloat16",
                "group_sizes": self.meta_group_sizes, "group_bit_widths": self.meta_group_bits,
                "qweight": self.qweight, "scales": self.scales, "qzeros": self.qzeros, "perm": self.perm,
                "qscale": self.qscale, "sscale": self.sscale, "scale_meta": self.scale_meta}

    @classmethod
    def from_packed_state(cls, state: dict, device) -> "QuantLinear":
        layer = cls(int(state["in_features"]), int(state["out_features"]), int(state["group_size"]), int(state["bits"]),
                    torch.float16 if state["dtype"] == "float16" else torch.bfloat16, state["group_sizes"],
                    state["group_bit_widths"], dense3=bool(state["dense3"]))
        words = 6 if layer.dense3 else 8
        g = layer.in_features // layer.group_size
        want = {"qweight": ((layer.in_features // 64, layer.out_features, words), torch.int32),
                "scales": ((g, layer.out_features), layer.dtype), "qzeros": ((g, layer.out_features), torch.uint8)}
        for name, (shape, dt) in want.items():
            t = state[name]
            if tuple(t.shape) != shape or t.dtype != dt:
                raise ValueError(f"packed state: {name} is {tuple(t.shape)} {t.dtype}, expected {shape} {dt}")
        if state["perm"] is not None and (state["perm"].dtype != torch.int32 or state["perm"].numel() != layer.in_features):
            raise ValueError("packed state: perm must be int32 of in_features entries")
        for name in ("qweight", "scales", "qzeros", "perm", "qscale", "sscale", "scale_meta"):
            t = state[name]
            setattr(layer, name, None if t is None else t.to(device).contiguous())
        return layer

    @torch.no_grad()
    def permute_input(self, x: torch.Tensor) -> torch.Tensor:
        """x[..., perm]: the activation gather of act-order.  Layers quantised against the same
        Hessian (q/k/v, gate/up) share ``perm``, so their caller can do this once and use
        ``forward_permuted`` on each of them."""
        if x.dtype != self.dtype:
            raise NotImplementedError(f"this QuantLinear takes {self.dtype} activations, got {x.dtype}")
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        if self.perm is not None:
            x2 = ops.permute_columns(x2, self.perm)
        return x2.reshape(shape)

    # From this many tokens per call on, the levels are expanded once into a transient dense weight and the
    # dense MFMA kernel multiplies by it (the packed kernel expands the same levels again for every 256-token
    # band).  Same product bit for bit; weights stay packed at rest.  0 = never.  GQ_WQ_EXPAND_MIN_TOKENS
    # in the environment sets the initial value.
    expand_min_tokens = int(os.environ.get("GQ_WQ_EXPAND_MIN_TOKENS", "2048"))

    def _expands(self, tokens: int) -> bool:
        least = self.expand_min_tokens
        if least <= 0:
            return False
        # a layer of fewer than 8192 rows takes twice as many tokens to win back the pass over its weights
        # (measured on Llama-2-7B shapes: from 2048 tokens at 11008 rows and more, from 4096 at 4096 rows)
        return tokens >= (least if self.out_features >= 8192 else 2 * least)

    @torch.no_grad()
    def forward_permuted(self, x_permuted: torch.Tensor) -> torch.Tensor:
        shape = x_permuted.shape
        x2 = x_permuted.reshape(-1, shape[-1])
        if x2.dtype != self.dtype:
            raise NotImplementedError(f"this QuantLinear takes {self.dtype} activations, got {x2.dtype}")
        bits = 3 if self.dense3 else 4
        if self._expands(x2.shape[0]):
            dense = ops.wq_expand(self.qweight, self.scales, self.qzeros, self.group_size, bits=bits)
            y = ops.gemm_nt(x2.contiguous(), dense)
        else:
            y = ops.w4a16_gemm(x2.contiguous(), self.qweight, self.scales, self.qzeros, self.group_size, bits=bits)
        return y.reshape(*shape[:-1], self.out_features)

    @classmethod
    @torch.no_grad()
    def concat(cls, layers) -> "QuantLinear":
        """One layer whose rows are the rows of ``layers`` in order (q/k/v, gate/up: Linears that read the
        same input, quantised against the same Hessian): one activation gather and one GEMM launch with
        three times the tiles instead of three.  The layers must agree in input width, act-order, fine
        group size, storage and type; ``forward`` returns the outputs side by side, each bit for bit what
        the layer alone returns (split with ``Tensor.split([l.out_features for l in layers], -1)``)."""
        layers = list(layers)
        first = layers[0]
        for other in layers[1:]:
            if (other.in_features != first.in_features or other.group_size != first.group_size or other.dense3 != first.dense3
                    or other.dtype != first.dtype or (other.perm is None) != (first.perm is None)
                    or (first.perm is not None and not torch.equal(other.perm, first.perm))):
                raise ValueError("QuantLinear.concat: layers differ in input width, act-order, group size, storage or type")
        out = cls(first.in_features, sum(l.out_features for l in layers), first.group_size, max(l.bits for l in layers),
                  first.dtype, dense3=first.dense3)
        # fine-grid description only: a concatenation is a serving object, unpack() of its parts gives the metadata back
        out.qweight = torch.cat([l.qweight for l in layers], dim=1).contiguous()
        out.scales = torch.cat([l.scales for l in layers], dim=1).contiguous()
        out.qzeros = torch.cat([l.qzeros for l in layers], dim=1).contiguous()
        out.perm = first.perm
        return out

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_permuted(self.permute_input(x))
