"""Tensor-level wrappers over the C ABI (include/gptq_amd.h).

PyTorch is used for device memory and streams only; every function validates its arguments
(device, dtype, contiguity, shape) and then hands raw device pointers to libgptq_amd.so.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib

GQ_F16, GQ_BF16 = 0, 1
SEM_KERNEL, SEM_PYTHON = 0, 1
PACK_STREAM, PACK_W4_PAIRS, PACK_W4_KBLOCK, PACK_W3_KBLOCK = 0, 1, 2, 3

_DTYPE_CODE = {torch.float16: GQ_F16, torch.bfloat16: GQ_BF16}


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise NotImplementedError(f"dtype {t.dtype}: only float16 and bfloat16 are supported") from None


def _check(t: torch.Tensor, name: str, dtype=None, ndim=None) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if t.device.type != "cuda":
        raise ValueError(f"{name}: must live on the GPU (got {t.device}); there is no CPU path")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dimensions, got {t.dim()}")


def _stream(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _int_array(vals: Sequence[int]):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def xtx_accumulate(hessian: torch.Tensor, x: torch.Tensor, save_lower_only: bool = False,
                   compute_lower_only: bool = True) -> None:
    """hessian (C, C) fp32 += x^T x for x (N, C) fp16/bf16, in place."""
    _check(hessian, "hessian", torch.float32, 2)
    _check(x, "x", None, 2)
    code = _dtype_code(x)
    n, c = x.shape
    if hessian.shape != (c, c):
        raise ValueError(f"hessian is {tuple(hessian.shape)}, x has {c} columns")
    if x.device != hessian.device:
        raise ValueError("hessian and x are on different devices")
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_xtx_accumulate(_p(hessian), _p(x), code, n, c, int(save_lower_only),
                                           int(compute_lower_only), _stream(x))
    _lib.check(rc, "gq_xtx_accumulate")


def xtx_split_hint(hidden: int, n_tokens: int, compute_lower_only: bool = True) -> int:
    """Number of partial Hessians that pays for calls of ``n_tokens`` tokens (1: do not split)."""
    return int(_lib.load().gq_xtx_split_hint(int(hidden), int(n_tokens), int(compute_lower_only)))


def xtx_accumulate_split(parts: torch.Tensor, x: torch.Tensor, save_lower_only: bool = False,
                         compute_lower_only: bool = True) -> None:
    """parts (S, C, C) fp32: the tokens of x (N, C) fp16/bf16 are cut into S runs, run p adds its X^T X
    into parts[p].  The Hessian is parts.sum(0), which ``xtx_fold`` leaves in parts[0]."""
    _check(parts, "parts", torch.float32, 3)
    _check(x, "x", None, 2)
    code = _dtype_code(x)
    n, c = x.shape
    if parts.shape[1:] != (c, c):
        raise ValueError(f"parts is {tuple(parts.shape)}, x has {c} columns")
    if x.device != parts.device:
        raise ValueError("parts and x are on different devices")
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_xtx_accumulate_split(_p(parts), parts.shape[0], c * c, _p(x), code, n, c, int(save_lower_only),
                                                 int(compute_lower_only), _stream(x))
    _lib.check(rc, "gq_xtx_accumulate_split")


def xtx_fold(parts: torch.Tensor) -> None:
    """parts[0] += parts[1] + parts[2] + ... (in this order), parts[1:] = 0."""
    _check(parts, "parts", torch.float32, 3)
    with torch.cuda.device(parts.device):
        rc = _lib.load().gq_xtx_fold(_p(parts), parts.shape[0], parts.shape[1] * parts.shape[2], parts.shape[1], _stream(parts))
    _lib.check(rc, "gq_xtx_fold")


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RMS normalisation over the last dimension of x (fp16 / bf16) in one pass, with the roundings of the usual
    operator chain: ``weight * (x.float() * rsqrt(mean(x.float() ** 2) + eps)).to(x.dtype)``."""
    _check(x, "x", None, None)
    _check(weight, "weight", x.dtype, 1)
    code = _dtype_code(x)
    hidden = x.shape[-1]
    if weight.shape[0] != hidden or weight.device != x.device:
        raise ValueError(f"weight is {tuple(weight.shape)} on {weight.device}, x has {hidden} columns on {x.device}")
    if out is None:
        out = torch.empty_like(x)
    else:
        _check(out, "out", x.dtype, None)
        if out.shape != x.shape or out.device != x.device:
            raise ValueError("out does not match x")
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_rmsnorm(_p(out), _p(x), _p(weight), code, x.numel() // hidden, hidden, float(eps), _stream(x))
    _lib.check(rc, "gq_rmsnorm")
    return out


def silu_mul(gate: torch.Tensor, up: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu(gate) * up in one pass (fp16 / bf16), rounded as ``F.silu(gate) * up`` rounds."""
    _check(gate, "gate", None, None)
    _check(up, "up", gate.dtype, None)
    code = _dtype_code(gate)
    if up.shape != gate.shape or up.device != gate.device:
        raise ValueError("gate and up differ in shape or device")
    if out is None:
        out = torch.empty_like(gate)
    else:
        _check(out, "out", gate.dtype, None)
        if out.shape != gate.shape or out.device != gate.device:
            raise ValueError("out does not match gate")
    with torch.cuda.device(gate.device):
        rc = _lib.load().gq_silu_mul(_p(out), _p(gate), _p(up), code, gate.numel(), _stream(gate))
    _lib.check(rc, "gq_silu_mul")
    return out


def rope_inplace(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_heads: int, head_dim: int) -> torch.Tensor:
    """Rotary embedding, in place, of the first ``n_heads * head_dim`` columns of x (batch, seq, columns) -- e.g. the q and k
    parts of a merged q/k/v product -- with cos / sin (batch or 1, seq, head_dim) of x's type, rounded operator by
    operator as ``x * cos + rotate_half(x) * sin`` is."""
    _check(x, "x", None, 3)
    _check(cos, "cos", x.dtype, 3)
    _check(sin, "sin", x.dtype, 3)
    code = _dtype_code(x)
    b, s, cols = x.shape
    if n_heads * head_dim > cols:
        raise ValueError(f"{n_heads} heads of {head_dim} do not fit {cols} columns")
    if cos.shape != sin.shape or cos.shape[1:] != (s, head_dim) or cos.shape[0] not in (1, b) or cos.device != x.device or sin.device != x.device:
        raise ValueError(f"cos / sin are {tuple(cos.shape)} / {tuple(sin.shape)}, expected ({b} or 1, {s}, {head_dim}) on {x.device}")
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_rope_inplace(_p(x), cols, _p(cos), _p(sin), 0 if cos.shape[0] == 1 else s * head_dim, b, s, int(n_heads),
                                         int(head_dim), code, _stream(x))
    _lib.check(rc, "gq_rope_inplace")
    return x


def xtx_tail_workspace(hidden: int) -> int:
    """Bytes of the tail buffers ``xtx_accumulate_tail`` needs for this hidden size (0: the size has no tail split)."""
    return int(_lib.load().gq_xtx_tail_workspace(int(hidden)))


def xtx_accumulate_tail(hessian: torch.Tensor, tail_ws: torch.Tensor, x: torch.Tensor) -> None:
    """hessian (C, C) fp32 += x^T x with the tiles of the partly filled last round summed in ``tail_ws`` (fp32,
    ``xtx_tail_workspace(C)`` bytes, zeroed by the caller once); ``xtx_fold_tail`` completes H."""
    _check(hessian, "hessian", torch.float32, 2)
    _check(tail_ws, "tail_ws", torch.float32, 1)
    _check(x, "x", None, 2)
    code = _dtype_code(x)
    n, c = x.shape
    if hessian.shape != (c, c):
        raise ValueError(f"hessian is {tuple(hessian.shape)}, x has {c} columns")
    if x.device != hessian.device or tail_ws.device != hessian.device:
        raise ValueError("hessian, tail_ws and x are on different devices")
    need = xtx_tail_workspace(c)
    if need == 0 or tail_ws.numel() * 4 < need:
        raise ValueError(f"tail_ws holds {tail_ws.numel() * 4} bytes, hidden size {c} needs {need} (0: no tail split)")
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_xtx_accumulate_tail(_p(hessian), _p(tail_ws), _p(x), code, n, c, _stream(x))
    _lib.check(rc, "gq_xtx_accumulate_tail")


def xtx_fold_tail(hessian: torch.Tensor, tail_ws: torch.Tensor) -> None:
    """Adds the tail buffers into ``hessian`` (index order), clears them, writes the tail tiles' mirror images."""
    _check(hessian, "hessian", torch.float32, 2)
    _check(tail_ws, "tail_ws", torch.float32, 1)
    c = hessian.shape[0]
    need = xtx_tail_workspace(c)
    if need == 0 or tail_ws.numel() * 4 < need:
        raise ValueError(f"tail_ws holds {tail_ws.numel() * 4} bytes, hidden size {c} needs {need} (0: no tail split)")
    with torch.cuda.device(hessian.device):
        rc = _lib.load().gq_xtx_fold_tail(_p(hessian), _p(tail_ws), c, _stream(hessian))
    _lib.check(rc, "gq_xtx_fold_tail")


def xtx_accumulate_f32(hessian: torch.Tensor, x: torch.Tensor) -> None:
    """hessian (C, C) fp32 += x^T x for x (N, C) fp32 (the reference's add_batch(use_kernel=False)
    arithmetic, gptq_py.py:31-33).  N is padded to a multiple of 4 with zero rows."""
    _check(hessian, "hessian", torch.float32, 2)
    _check(x, "x", torch.float32, 2)
    n, c = x.shape
    if hessian.shape != (c, c):
        raise ValueError(f"hessian is {tuple(hessian.shape)}, x has {c} columns")
    if x.device != hessian.device:
        raise ValueError("hessian and x are on different devices")
    if c % 4:
        raise NotImplementedError("fp32 activations: the hidden size must be a multiple of 4")
    if n % 4:
        x = torch.cat([x, x.new_zeros(4 - n % 4, c)], dim=0)
        n = x.shape[0]
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_xtx_accumulate_f32(_p(hessian), _p(x), n, c, _stream(x))
    _lib.check(rc, "gq_xtx_accumulate_f32")


def hessian_invert(hessian: torch.Tensor, n_samples: float, damp_ratio: float = 1e-2, act_order: bool = True,
                   max_try: int = 100) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor], int]:
    """In place on ``hessian`` (scaled, permuted, damped as the reference leaves it).
    Returns (hessian_inv U, perm int32 | None, perm_inv int32 | None, tries)."""
    _check(hessian, "hessian", torch.float32, 2)
    c = hessian.shape[0]
    if hessian.shape != (c, c):
        raise ValueError("hessian must be square")
    dev = hessian.device
    lib = _lib.load()
    u = torch.empty_like(hessian)
    perm = torch.empty(c, dtype=torch.int32, device=dev) if act_order else None
    perm_inv = torch.empty(c, dtype=torch.int32, device=dev) if act_order else None
    ws_bytes = lib.gq_hessian_invert_workspace(c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    tries = ctypes.c_int(0)
    with torch.cuda.device(dev):
        rc = lib.gq_hessian_invert(_p(hessian), _p(u), _p(perm), _p(perm_inv), c, float(n_samples), float(damp_ratio),
                                   int(act_order), int(max_try), _p(ws), ws_bytes, ctypes.byref(tries), _stream(hessian))
    _lib.check(rc, "gq_hessian_invert")
    return u, perm, perm_inv, tries.value


class PendingInversion:
    """What ``hessian_invert_begin`` leaves behind: the outputs (valid in stream order if the first try
    succeeds), the verdict of that try in device memory and the workspace ``hessian_invert_finish`` needs."""

    def __init__(self, hessian, u, perm, perm_inv, workspace, verdict):
        self.hessian, self.u, self.perm, self.perm_inv = hessian, u, perm, perm_inv
        self.workspace, self.verdict = workspace, verdict


def hessian_invert_begin(hessian: torch.Tensor, n_samples: float, damp_ratio: float = 1e-2,
                         act_order: bool = True) -> PendingInversion:
    """``hessian_invert`` up to and including its first try, without waiting for the device."""
    _check(hessian, "hessian", torch.float32, 2)
    c = hessian.shape[0]
    if hessian.shape != (c, c):
        raise ValueError("hessian must be square")
    dev = hessian.device
    lib = _lib.load()
    u = torch.empty_like(hessian)
    perm = torch.empty(c, dtype=torch.int32, device=dev) if act_order else None
    perm_inv = torch.empty(c, dtype=torch.int32, device=dev) if act_order else None
    ws_bytes = lib.gq_hessian_invert_workspace(c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    verdict = torch.empty(2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gq_hessian_invert_begin(_p(hessian), _p(u), _p(perm), _p(perm_inv), c, float(n_samples),
                                         float(damp_ratio), int(act_order), _p(ws), ws_bytes, _p(verdict),
                                         _stream(hessian))
    _lib.check(rc, "gq_hessian_invert_begin")
    return PendingInversion(hessian, u, perm, perm_inv, ws, verdict)


def hessian_invert_finish(pending: PendingInversion, max_try: int = 100) -> int:
    """Reads the verdict of the first try (waits for the device) and makes the remaining tries if it
    failed.  Returns the number of tries; more than 1 means ``pending.u`` was written again and whatever
    was computed from it in between is void."""
    v = pending.verdict.tolist()
    verdict = (ctypes.c_int32 * 2)(int(v[0]), int(v[1]))
    tries = ctypes.c_int(0)
    h = pending.hessian
    with torch.cuda.device(h.device):
        rc = _lib.load().gq_hessian_invert_finish(_p(h), _p(pending.u), h.shape[0], int(max_try), _p(pending.workspace),
                                                  pending.workspace.numel(), verdict, ctypes.byref(tries), _stream(h))
    _lib.check(rc, "gq_hessian_invert_finish")
    return tries.value


def find_params(weights_t: torch.Tensor, col0: int, ncols: int, bits: int, sym: bool = False):
    """weights_t (C, R) fp32 -> scale (R,), qzero (R,) fp32 over columns [col0, col0+ncols)."""
    _check(weights_t, "weights_t", torch.float32, 2)
    c, r = weights_t.shape
    scale = torch.empty(r, dtype=torch.float32, device=weights_t.device)
    qzero = torch.empty_like(scale)
    with torch.cuda.device(weights_t.device):
        rc = _lib.load().gq_find_params(_p(weights_t), c, r, col0, ncols, bits, int(sym), _p(scale), _p(qzero),
                                        _stream(weights_t))
    _lib.check(rc, "gq_find_params")
    return scale, qzero


def quantize_range(quant, scale, out_q, qzero, maxq, hessian_inv, weights, error, a, b) -> None:
    """The reference's native op, same argument order (package/gptq/__init__.py:16-28)."""
    for t, name in ((quant, "quant"), (scale, "scale"), (qzero, "qzero"), (hessian_inv, "hessian_inv"),
                    (weights, "weights"), (error, "error")):
        _check(t, name, torch.float32)
    c, r = weights.shape
    if hessian_inv.shape[1] != c:
        raise RuntimeError("H shape mismatch")  # the reference's only check (gptq_cpp.cpp:49)
    if quant.shape != (c, r) or error.shape != (c, r) or scale.shape != (r,) or qzero.shape != (r,):
        raise ValueError("quantize_range: inconsistent shapes")
    q_ptr = None
    if out_q is not None and out_q.device.type != "meta":  # a meta tensor means "no integer output"
        _check(out_q, "out_q", torch.int16)
        q_ptr = out_q
    with torch.cuda.device(weights.device):
        rc = _lib.load().gq_quantize_range(_p(quant), _p(scale), _p(q_ptr), _p(qzero), float(maxq), _p(hessian_inv),
                                           _p(weights), _p(error), c, r, int(a), int(b), _stream(weights))
    _lib.check(rc, "gq_quantize_range")


def quantize_layer(weight: torch.Tensor, perm: Optional[torch.Tensor], hessian_inv: Optional[torch.Tensor],
                   group_sizes: Sequence[int], group_bits: Sequence[int], sym: bool = False,
                   semantics: int = SEM_KERNEL, want_quant: bool = True, want_error: bool = True,
                   block_sizes: Optional[Sequence[int]] = None, scale_bits: int = 0,
                   shrink: Optional[torch.Tensor] = None, norm: float = 2.4):
    """weight (R, C) fp16/bf16 -> dict(qweight u8 (R,C), scale (R,G) weight dtype, qzero u8 (R,G),
    qscale u8 (R,G) | None, sscale (G,) weight dtype | None, quant_t fp32 (C,R) | None,
    error_t fp32 (C,R) | None).  hessian_inv None = round to nearest.
    block_sizes: the PyTorch loop's blocks (None = the groups); scale_bits > 0: EXL2 scale
    quantisation; shrink: device tensor of shrink candidates, switches the error-minimising search on."""
    _check(weight, "weight", None, 2)
    code = _dtype_code(weight)
    r, c = weight.shape
    dev = weight.device
    g = len(group_sizes)
    if len(group_bits) != g:
        raise ValueError("group_sizes and group_bit_widths differ in length")
    lib = _lib.load()
    qweight = torch.empty(r, c, dtype=torch.uint8, device=dev)
    scale = torch.empty(r, g, dtype=weight.dtype, device=dev)
    qzero = torch.empty(r, g, dtype=torch.uint8, device=dev)
    qscale = torch.empty(r, g, dtype=torch.uint8, device=dev) if scale_bits > 0 else None
    sscale = torch.empty(g, dtype=weight.dtype, device=dev) if scale_bits > 0 else None
    quant_t = torch.empty(c, r, dtype=torch.float32, device=dev) if want_quant else None
    error_t = None
    if hessian_inv is not None:
        _check(hessian_inv, "hessian_inv", torch.float32, 2)
        if hessian_inv.shape != (c, c):
            raise RuntimeError("H shape mismatch")
        if perm is not None:
            _check(perm, "perm", torch.int32, 1)
        error_t = torch.empty(c, r, dtype=torch.float32, device=dev) if want_error else None
    if shrink is not None:
        _check(shrink, "shrink", torch.float32, 1)
    ws_bytes = lib.gq_quantize_layer_workspace(r, c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    gs, gb = _int_array(group_sizes), _int_array(group_bits)
    extended = block_sizes is not None or scale_bits > 0 or shrink is not None
    with torch.cuda.device(dev):
        if extended:
            opt = _lib.GridOptions(int(sym), int(scale_bits), int(shrink is not None),
                                   shrink.data_ptr() if shrink is not None else None,
                                   shrink.numel() if shrink is not None else 0, float(norm))
            bsz = _int_array(block_sizes) if block_sizes is not None else None
            rc = lib.gq_quantize_layer_ex(_p(weight), code, _p(perm), _p(hessian_inv), r, c, gs, gb, g, bsz,
                                          len(block_sizes) if block_sizes is not None else 0, int(semantics),
                                          ctypes.byref(opt), _p(qweight), _p(scale), _p(qzero), _p(qscale), _p(sscale),
                                          _p(quant_t), _p(error_t), _p(ws), ws_bytes, _stream(weight))
            _lib.check(rc, "gq_quantize_layer_ex")
        elif hessian_inv is None:
            rc = lib.gq_quantize_rtn(_p(weight), code, r, c, gs, gb, g, int(sym), _p(qweight), _p(scale), _p(qzero),
                                     _p(quant_t), _p(ws), ws_bytes, _stream(weight))
            _lib.check(rc, "gq_quantize_rtn")
        else:
            rc = lib.gq_quantize_layer(_p(weight), code, _p(perm), _p(hessian_inv), r, c, gs, gb, g, int(sym),
                                       int(semantics), _p(qweight), _p(scale), _p(qzero), _p(quant_t), _p(error_t),
                                       _p(ws), ws_bytes, _stream(weight))
            _lib.check(rc, "gq_quantize_layer")
    return {"qweight": qweight, "scale": scale, "qzero": qzero, "qscale": qscale, "sscale": sscale,
            "quant_t": quant_t, "error_t": error_t}


def quant_metrics(weight, perm, hessian_inv, quant_t, error_t, perm_inv=None) -> torch.Tensor:
    """Returns the four sums (double, device) behind gptq_quant's metrics.  ``perm_inv`` (int32, the inverse
    of ``perm``) is optional and only makes the call faster."""
    _check(weight, "weight", None, 2)
    r, c = weight.shape
    if perm_inv is not None:
        _check(perm_inv, "perm_inv", torch.int32, 1)
        if perm is None or perm_inv.numel() != c:
            raise ValueError("quant_metrics: perm_inv needs perm and one entry per column")
    sums = torch.empty(4, dtype=torch.float64, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _lib.load().gq_quant_metrics(_p(weight), _dtype_code(weight), _p(perm), _p(perm_inv), _p(hessian_inv),
                                          _p(quant_t), _p(error_t), r, c, _p(sums), _stream(weight))
    _lib.check(rc, "gq_quant_metrics")
    return sums


def dequantize(qweight: torch.Tensor, scale: torch.Tensor, qzero: torch.Tensor, perm_inv: Optional[torch.Tensor],
               group_sizes: Sequence[int]) -> torch.Tensor:
    """Dense (R, C) weight in scale's dtype, columns un-permuted (reconstruct_nn_linear's weight)."""
    _check(qweight, "qweight", torch.uint8, 2)
    _check(scale, "scale", None, 2)
    _check(qzero, "qzero", torch.uint8, 2)
    r, c = qweight.shape
    g = scale.shape[1]
    if scale.shape != (r, g) or qzero.shape != (r, g) or len(group_sizes) != g or sum(group_sizes) != c:
        raise ValueError("dequantize: inconsistent shapes")
    code = _dtype_code(scale)
    dev = qweight.device
    if perm_inv is not None:
        _check(perm_inv, "perm_inv", torch.int32, 1)
    uniform = all(s == group_sizes[0] for s in group_sizes)
    col_group = None
    if not uniform:
        col_group = torch.repeat_interleave(torch.arange(g, dtype=torch.int32, device=dev),
                                            torch.tensor(list(group_sizes), device=dev)).contiguous()
    out = torch.empty(r, c, dtype=scale.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().gq_dequantize(_p(qweight), _p(scale), _p(qzero), _p(perm_inv), r, c, g,
                                       int(group_sizes[0]) if uniform else 0, _p(col_group), code, _p(out),
                                       _stream(qweight))
    _lib.check(rc, "gq_dequantize")
    return out


def words_per_row(n: int, bits: int) -> int:
    return (n * bits + 31) // 32


def pack_rows(values: torch.Tensor, bits: int, layout: int = PACK_STREAM) -> torch.Tensor:
    _check(values, "values", torch.uint8, 2)
    rows, n = values.shape
    if layout == PACK_W4_KBLOCK:
        if bits != 4 or n % 64:
            raise ValueError("pack_rows: the k-block layout needs 4-bit values and n % 64 == 0")
        out = torch.empty(n // 64, rows, 8, dtype=torch.int32, device=values.device)
    elif layout == PACK_W3_KBLOCK:
        if bits != 3 or n % 64:
            raise ValueError("pack_rows: the 3-bit k-block layout needs 3-bit values and n % 64 == 0")
        out = torch.empty(n // 64, rows, 6, dtype=torch.int32, device=values.device)
    else:
        out = torch.empty(rows, words_per_row(n, bits), dtype=torch.int32, device=values.device)
    with torch.cuda.device(values.device):
        rc = _lib.load().gq_pack_rows(_p(values), rows, n, bits, layout, _p(out), _stream(values))
    _lib.check(rc, "gq_pack_rows")
    return out


def unpack_rows(packed: torch.Tensor, bits: int, n: int, layout: int = PACK_STREAM) -> torch.Tensor:
    if layout in (PACK_W4_KBLOCK, PACK_W3_KBLOCK):
        _check(packed, "packed", torch.int32, 3)
        rows = packed.shape[1]
        want_bits, words = (4, 8) if layout == PACK_W4_KBLOCK else (3, 6)
        if bits != want_bits or n % 64 or packed.shape[0] != n // 64 or packed.shape[2] != words:
            raise ValueError(f"unpack_rows: k-block layout is (n/64, rows, {words}) words of {want_bits}-bit values")
    else:
        _check(packed, "packed", torch.int32, 2)
        rows = packed.shape[0]
        if packed.shape[1] != words_per_row(n, bits):
            raise ValueError("unpack_rows: word count does not match n and bits")
    out = torch.empty(rows, n, dtype=torch.uint8, device=packed.device)
    with torch.cuda.device(packed.device):
        rc = _lib.load().gq_unpack_rows(_p(packed), rows, n, bits, layout, _p(out), _stream(packed))
    _lib.check(rc, "gq_unpack_rows")
    return out


def pack_lower(hessian: torch.Tensor, flat: torch.Tensor) -> None:
    """flat[r (r + 1) / 2 + c] = hessian[r][c] for c <= r (fp32, C x C); flat has at least C (C + 1) / 2 elements."""
    _check(hessian, "hessian", torch.float32, 2)
    _check(flat, "flat", torch.float32, 1)
    c = hessian.shape[0]
    if hessian.shape[1] != c or flat.numel() < c * (c + 1) // 2:
        raise ValueError("pack_lower: a square matrix and a buffer of C (C + 1) / 2 elements")
    with torch.cuda.device(hessian.device):
        rc = _lib.load().gq_pack_lower(_p(hessian), c, _p(flat), _stream(hessian))
    _lib.check(rc, "gq_pack_lower")


def unpack_lower(flat: torch.Tensor, hessian: torch.Tensor) -> None:
    """Inverse of pack_lower; the upper triangle is written as the mirror image."""
    _check(hessian, "hessian", torch.float32, 2)
    _check(flat, "flat", torch.float32, 1)
    c = hessian.shape[0]
    if hessian.shape[1] != c or flat.numel() < c * (c + 1) // 2:
        raise ValueError("unpack_lower: a square matrix and a buffer of C (C + 1) / 2 elements")
    with torch.cuda.device(hessian.device):
        rc = _lib.load().gq_unpack_lower(_p(flat), c, _p(hessian), _stream(hessian))
    _lib.check(rc, "gq_unpack_lower")


def permute_columns(x: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    _check(x, "x", None, 2)
    _check(perm, "perm", torch.int32, 1)
    if x.element_size() != 2:
        raise NotImplementedError("permute_columns: 16-bit elements only")
    m, c = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_permute_columns(_p(x), _p(perm), m, c, _p(out), _stream(x))
    _lib.check(rc, "gq_permute_columns")
    return out


def w4a16_gemm(x: torch.Tensor, qweight_packed: torch.Tensor, scale_t: torch.Tensor, qzero_t: torch.Tensor,
               group_size: int, bits: int = 4) -> torch.Tensor:
    """x (M, C) fp16 or bf16, columns in permuted order -> y (M, R) of the same type.
    qweight_packed (C/64, R, 8) int32 in the 4-bit k-block layout, or (C/64, R, 6) in the dense 3-bit one
    (``bits`` = 3); scale_t (G, R) of x's type, qzero_t (G, R) uint8."""
    _check(x, "x", None, 2)
    code = _dtype_code(x)
    _check(qweight_packed, "qweight_packed", torch.int32, 3)
    _check(scale_t, "scale_t", x.dtype, 2)
    _check(qzero_t, "qzero_t", torch.uint8, 2)
    scale, qzero = scale_t, qzero_t
    m, c = x.shape
    r = qweight_packed.shape[1]
    if bits not in (3, 4):
        raise ValueError("w4a16_gemm: bits must be 3 or 4")
    if (c % 64 or qweight_packed.shape[0] * 64 != c or qweight_packed.shape[2] != (8 if bits == 4 else 6) or group_size <= 0
            or c % group_size or scale.shape != (c // group_size, r) or qzero.shape != scale.shape):
        raise ValueError("w4a16_gemm: inconsistent shapes")
    y = torch.empty(m, r, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_wq_gemm(_p(y), _p(x), _p(qweight_packed), _p(scale), _p(qzero), m, r, c, int(group_size),
                                    int(bits), code, _stream(x))
    _lib.check(rc, "gq_wq_gemm")
    return y

def _check_packed(qweight_packed, scale_t, qzero_t, group_size, bits, what):
    _check(qweight_packed, "qweight_packed", torch.int32, 3)
    _check(scale_t, "scale_t", None, 2)
    _check(qzero_t, "qzero_t", torch.uint8, 2)
    if bits not in (3, 4):
        raise ValueError(f"{what}: bits must be 3 or 4")
    c, r = qweight_packed.shape[0] * 64, qweight_packed.shape[1]
    if (qweight_packed.shape[2] != (8 if bits == 4 else 6) or group_size <= 0 or c % group_size
            or scale_t.shape != (c // group_size, r) or qzero_t.shape != scale_t.shape):
        raise ValueError(f"{what}: inconsistent shapes")
    return r, c


def wq_expand(qweight_packed: torch.Tensor, scale_t: torch.Tensor, qzero_t: torch.Tensor, group_size: int,
              bits: int = 4, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense (R, C) weight of scale_t's type from a packed operand, columns in the operand's own order:
    the values ``w4a16_gemm`` multiplies by, written out once."""
    r, c = _check_packed(qweight_packed, scale_t, qzero_t, group_size, bits, "wq_expand")
    code = _dtype_code(scale_t)
    if out is None:
        out = torch.empty(r, c, dtype=scale_t.dtype, device=scale_t.device)
    elif out.shape != (r, c) or out.dtype != scale_t.dtype or not out.is_contiguous() or out.device != scale_t.device:
        raise ValueError("wq_expand: out must be a contiguous (R, C) tensor of the scales' type on their device")
    with torch.cuda.device(scale_t.device):
        rc = _lib.load().gq_wq_expand(_p(out), _p(qweight_packed), _p(scale_t), _p(qzero_t), r, c, int(group_size), int(bits),
                                      code, _stream(scale_t))
    _lib.check(rc, "gq_wq_expand")
    return out


def gemm_nt(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """x (M, K) @ weight (N, K)^T -> (M, N), fp16 or bf16, fp32 accumulation: the dense Linear forward
    of the calibration replay (llama.py:176-177) on the hand-written MFMA kernel."""
    _check(x, "x", None, 2)
    _check(weight, "weight", x.dtype, 2)
    code = _dtype_code(x)
    m, k = x.shape
    n = weight.shape[0]
    if weight.shape[1] != k:
        raise ValueError(f"gemm_nt: x has {k} columns, weight has {weight.shape[1]}")
    if k % 64:
        raise NotImplementedError("gemm_nt: in_features must be a multiple of 64")
    if x.device != weight.device:
        raise ValueError("gemm_nt: x and weight are on different devices")
    y = torch.empty(m, n, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().gq_gemm_nt(_p(y), _p(x), _p(weight), m, n, k, code, _stream(x))
    _lib.check(rc, "gq_gemm_nt")
    return y
