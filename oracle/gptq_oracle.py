"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GPTQ hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product package never does.

It restates, on CPU tensors, what the reference's Python layer computes
(citations are ``file:line`` in the reference checkout):

  =====================  ==========================================================
  here                   reference
  =====================  ==========================================================
  OracleHessian          gptq_py.py:9-66     HessianHook.add_batch(use_kernel=False), invert
  find_params            quant.py:89-118     Quantizer.find_params (min/max, symmetric and EXL2 grids)
  quantize / dequantize  quant.py:10-30      (+ quantize2 / dequantize2, quant.py:40-69)
  mse_refine             quant.py:120-165    Quantizer.mse, PyTorch branch
  quantize_range_kernel  package/gptq/gptq_cpp.cpp:33-84 + gptq_cpp_kernel.cu:21-137
                         (through the C restatement ``kernel_path_ref.c``)
  gptq_quant             gptq_py.py:69-251   all three branches + result collation
  construct_weight       quant.py:222-237, 262-279  construct_matrix / reconstruct_nn_linear
  pack_rows / unpack_rows, w4a16_linear : no reference counterpart (the reference never
                         packs and runs a dense fp16 GEMM, llama.py:172-177); these
                         DEFINE the packed layout and the result the HIP kernels must match.
  =====================  ==========================================================

Pinning: the Python-path functions are checked against fixtures produced by importing the
reference itself (tests/golden/make_golden.py).  The kernel-path restatement has no golden
vectors in the reference (its native code needs nvcc + CUTLASS, neither is present), so its
parity is pinned only by (a) the reference's own Python glue run around it, (b) identities
the reference implies (first-group grid equals the Python path's, ``quant`` equals the
reconstructed weight bit for bit); beyond that it is "parity unpinned".
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np
import torch

EPSILON = 1e-12  # quant.py:7

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[ctypes.CDLL] = None


def _lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.gqref_quantize_range.restype = ctypes.c_int
        lib.gqref_quantize_range.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ]
        lib.gqref_xtx_accumulate.restype = ctypes.c_int
        lib.gqref_xtx_accumulate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int64, ctypes.c_int64]
        lib.gqref_f2h.restype = ctypes.c_uint16
        lib.gqref_f2h.argtypes = [ctypes.c_float]
        lib.gqref_h2f.restype = ctypes.c_float
        lib.gqref_h2f.argtypes = [ctypes.c_uint16]
        lib.gqref_hsub.restype = ctypes.c_uint16
        lib.gqref_hsub.argtypes = [ctypes.c_uint16, ctypes.c_uint16]
        lib.gqref_hmul.restype = ctypes.c_uint16
        lib.gqref_hmul.argtypes = [ctypes.c_uint16, ctypes.c_uint16]
        _LIB = lib
    return _LIB


# ---------------------------------------------------------------------------------------
# Hessian accumulation and inversion (gptq_py.py:9-66)
# ---------------------------------------------------------------------------------------

class OracleHessian:
    """State machine with the same fields as the reference's HessianHook."""

    def __init__(self) -> None:
        self.hessian: Optional[torch.Tensor] = None
        self.n_samples = 0
        self.hessian_inv: Optional[torch.Tensor] = None
        self.perm: Optional[torch.Tensor] = None
        self.perm_inv: Optional[torch.Tensor] = None
        self.tries = 0

    @torch.no_grad()
    def add_batch(self, inp: torch.Tensor) -> None:
        # gptq_py.py:25-33: a 2-D batch counts as ONE sample; n_samples counts sequences
        assert self.hessian_inv is None
        x = inp if inp.dim() > 2 else inp.unsqueeze(0)
        self.n_samples += x.shape[0]
        d = x.shape[-1]
        if self.hessian is None:
            self.hessian = torch.zeros(d, d, dtype=torch.float32)
        x2 = x.reshape(-1, d).to(torch.float32)
        torch.addmm(self.hessian, x2.t(), x2, beta=1, alpha=1, out=self.hessian)

    @torch.no_grad()
    def invert(self, damp_ratio: float = 1e-2, act_order: bool = True, max_try: int = 100) -> torch.Tensor:
        # gptq_py.py:41-44
        self.hessian *= 2.0 / self.n_samples
        diag = self.hessian.diagonal()
        diag[diag == 0.0] = 1.0
        # gptq_py.py:46-49
        if act_order:
            self.perm = torch.argsort(self.hessian.diag(), descending=True)
            self.hessian = self.hessian[self.perm][:, self.perm].contiguous()
            self.perm_inv = torch.argsort(self.perm)
        # gptq_py.py:52: damp is computed once, after permutation, and re-added per try
        damp = damp_ratio * self.hessian.diag().mean()  # .diag(): contiguous copy, same reduction order
        self.tries = 0
        while self.tries < max_try:
            self.tries += 1
            self.hessian.diagonal().add_(damp)
            low, _ = torch.linalg.cholesky_ex(self.hessian, upper=False, check_errors=False)
            inv = torch.cholesky_inverse(low, upper=False)
            up, _ = torch.linalg.cholesky_ex(inv, upper=True, check_errors=False)
            self.hessian_inv = up
            if not torch.isnan(up).any():
                return up
        raise AssertionError("Cholesky kept producing NaN")  # gptq_py.py:65


def upper_inverse_factor_f64(h_damped: torch.Tensor) -> torch.Tensor:
    """fp64 value of the factor the reference computes in fp32 at gptq_py.py:59-61:
    U upper triangular with (H)^-1 = U^T U.  Used to judge fp32 implementations."""
    h = h_damped.to(torch.float64)
    low = torch.linalg.cholesky(h)
    inv = torch.cholesky_inverse(low)
    return torch.linalg.cholesky(inv, upper=True)


# ---------------------------------------------------------------------------------------
# Quantization grid (quant.py:10-30, 89-118)
# ---------------------------------------------------------------------------------------

def quantize(x, scale, qzero, maxq):
    return torch.clamp(torch.round(x / scale) + qzero, 0.0, maxq)  # quant.py:19


def dequantize(qx, scale, qzero):
    return (qx - qzero) * scale  # quant.py:30


def quantize2(x, scale, maxq):
    return torch.clamp(torch.round(torch.sqrt(x / scale)), 1.0, maxq + 1.0)  # quant.py:51


def dequantize2(qx, scale):
    return qx * qx * scale  # quant.py:62


class Grid:
    """One group's grid: scale, qzero (R, 1), maxq; EXL2: qscale (R, 1), sscale (1, 1), smaxq."""

    def __init__(self):
        self.scale = self.qzero = self.maxq = self.qscale = self.sscale = self.smaxq = None


def find_params(x: torch.Tensor, bit_width, sym: bool = False, scale_bit_width=None, full: bool = False):
    """x (..., R, C) -> scale, qzero of shape (..., R, 1), maxq scalar (or the whole Grid with
    ``full``).  quant.py:99-118; every operand fp32 as in the reference (bit widths are tensors
    there, so 2 ** bit_width is an fp32 tensor)."""
    g = Grid()
    exl2 = scale_bit_width is not None and int(scale_bit_width) > 0
    g.maxq = torch.tensor(2.0, dtype=torch.float32) ** int(bit_width) - 1.0
    if sym or exl2:
        g.scale = x.abs().amax(dim=-1, keepdim=True) * (2.0 / g.maxq) + EPSILON
        g.qzero = torch.full_like(g.scale, float(torch.round((g.maxq + 1.0) * 0.5)))
        if exl2:
            g.smaxq = torch.tensor(2.0, dtype=torch.float32) ** int(scale_bit_width) - 1.0
            g.sscale = g.scale.amax(dim=-2, keepdim=True) / (g.smaxq + 1.0) ** 2.0 + EPSILON
            g.qscale = quantize2(g.scale, g.sscale, g.smaxq)
            g.scale = dequantize2(g.qscale, g.sscale)
    else:
        hi = torch.relu(x.amax(dim=-1, keepdim=True))
        lo = -torch.relu(-x.amin(dim=-1, keepdim=True))
        g.scale = (hi - lo) / g.maxq + EPSILON
        g.qzero = torch.round(-lo / g.scale)
    if full:
        return g
    return g.scale, g.qzero, float(g.maxq)


def shrink_candidates(max_shrink: float = 0.8, n_grid: int = 100) -> torch.Tensor:
    return 1.0 - torch.arange(0.0, max_shrink, 1.0 / n_grid)  # quant.py:151


def _first_argmin(err: torch.Tensor) -> torch.Tensor:
    """index of the first minimum along the last dimension, keepdim"""
    best = err.min(dim=-1, keepdim=True).values
    q = err.shape[-1]
    idx = torch.where(err == best, torch.arange(q).expand(err.shape), torch.full(err.shape, q))
    return idx.min(dim=-1, keepdim=True).values


def mse_refine(g: Grid, x: torch.Tensor, max_shrink: float = 0.8, n_grid: int = 100, norm: float = 2.4,
               exact: bool = False) -> None:
    """Quantizer.mse, PyTorch branch (quant.py:151-165), in place on ``g``.
    exact=False: the reference's arithmetic (fp32 powers summed by torch).
    exact=True: the same candidates with the powers and sums in float64 and the first minimum
    taken -- the definition the HIP kernel follows; it differs from the above only where two
    candidates' errors agree to fp32 rounding."""
    p = shrink_candidates(max_shrink, n_grid)

    def pick(cand, ref, dim):
        d = (cand - ref).abs()
        if exact:
            return _first_argmin(d.to(torch.float64).pow(norm).sum(dim=dim))
        return d.pow(norm).sum(dim=dim).argmin(dim=-1, keepdim=True)

    xq = x[..., None]
    sp = g.scale[..., None] * p
    cand = dequantize(quantize(xq, sp, g.qzero[..., None], g.maxq), sp, g.qzero[..., None])
    g.scale = g.scale * p[pick(cand, xq, -2)]
    if g.sscale is not None:
        g.sscale = g.scale.amax(dim=-2, keepdim=True) / (g.smaxq + 1.0) ** 2.0 + EPSILON
        sq = g.scale[..., None]
        ssp = g.sscale[..., None] * p
        cand = dequantize2(quantize2(sq, ssp, g.smaxq), ssp)
        g.sscale = g.sscale * p[pick(cand, sq, -3)]
        g.qscale = quantize2(g.scale, g.sscale, g.smaxq)
        g.scale = dequantize2(g.qscale, g.sscale)


# ---------------------------------------------------------------------------------------
# Native quantize loop (gptq_cpp.cpp:33-84) through the C restatement
# ---------------------------------------------------------------------------------------

def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def quantize_range_kernel(quant_t, scale, out_q, qzero, maxq, hessian_inv, weights_t, error_t, a, b):
    """Same argument order, layouts ((C, R) fp32; out_q int16 or None) and in-place effects as
    ``gptq.gptq_quantize_range`` (package/gptq/__init__.py:16-28)."""
    for t, dt in ((quant_t, torch.float32), (scale, torch.float32), (qzero, torch.float32),
                  (hessian_inv, torch.float32), (weights_t, torch.float32), (error_t, torch.float32)):
        assert t.dtype == dt and t.is_contiguous() and t.device.type == "cpu"
    if out_q is not None:
        assert out_q.dtype == torch.int16 and out_q.is_contiguous()
    rows, columns = weights_t.shape
    rc = _lib().gqref_quantize_range(_ptr(quant_t), _ptr(scale), _ptr(out_q), _ptr(qzero), float(maxq),
                                     _ptr(hessian_inv), int(hessian_inv.shape[1]), _ptr(weights_t),
                                     _ptr(error_t), int(rows), int(columns), int(a), int(b))
    if rc != 0:
        raise RuntimeError("H shape mismatch")  # gptq_cpp.cpp:49


def xtx_accumulate_exact(hessian: torch.Tensor, x: torch.Tensor) -> None:
    """H += X^T X with every element accumulated in double and rounded once."""
    assert hessian.dtype == torch.float32 and hessian.is_contiguous() and x.is_contiguous()
    dt = {torch.float16: 0, torch.bfloat16: 1}[x.dtype]
    n, c = x.shape
    rc = _lib().gqref_xtx_accumulate(_ptr(hessian), _ptr(x.view(torch.int16)), dt, n, c)
    assert rc == 0


# ---------------------------------------------------------------------------------------
# gptq_quant (gptq_py.py:69-251)
# ---------------------------------------------------------------------------------------

@torch.no_grad()
def gptq_quant(weight: torch.Tensor, hook: Optional[OracleHessian], group_sizes: torch.Tensor,
               group_bit_widths: torch.Tensor, use_kernel_semantics: bool = True,
               block_sizes: Optional[torch.Tensor] = None, symmetric: bool = False,
               scale_bit_width=None, mse: bool = False, max_shrink: float = 0.8, n_grid: int = 100,
               norm: float = 2.4, mse_exact: bool = False) -> dict:
    """Returns {'quant_meta': {...}, 'metrics': {...}, 'debug': {...}} with quant_meta and
    metrics as in gptq_py.py:240-251."""
    n_rows, n_cols = weight.shape
    dtype = weight.dtype
    w_ref = weight
    w = weight.to(torch.float32).clone()
    if hook is not None:
        if hook.perm is not None:
            w = w[:, hook.perm]
        dead = hook.hessian.diagonal() == 0.0  # gptq_py.py:95-96 (a no-op after invert)
        w[:, dead] = 0.0
    bounds = [0] + torch.cumsum(group_sizes, 0).tolist()
    grids = []
    debug: dict = {}

    def new_grid(cols: torch.Tensor, bw, search_cols: torch.Tensor) -> Grid:
        g = find_params(cols, bw, symmetric, scale_bit_width, full=True)
        if mse:
            mse_refine(g, search_cols, max_shrink, n_grid, norm, exact=mse_exact)
        grids.append(g)
        return g

    if hook is None:  # gptq_py.py:101-124 round to nearest
        quant = torch.empty_like(w)
        qweight = torch.empty_like(w)
        for bw, i1, i2 in zip(group_bit_widths.tolist(), bounds[:-1], bounds[1:]):
            g = new_grid(w[:, i1:i2], bw, w[:, i1:i2])
            qweight[:, i1:i2] = quantize(w[:, i1:i2], g.scale, g.qzero, g.maxq)
            quant[:, i1:i2] = dequantize(qweight[:, i1:i2], g.scale, g.qzero)
        error = torch.zeros((), dtype=torch.float32)
    elif use_kernel_semantics:  # gptq_py.py:125-157
        w_t = w.t().clone(memory_format=torch.contiguous_format)  # a copy even for one row (w stays unmodified)
        quant_t = torch.empty_like(w_t)
        error_t = torch.empty_like(w_t)
        q_t = torch.empty_like(w_t, dtype=torch.int16)
        u = hook.hessian_inv.contiguous()
        for bw, i1, i2 in zip(group_bit_widths.tolist(), bounds[:-1], bounds[1:]):
            g = new_grid(w[:, i1:i2], bw, w[:, i1:i2])  # grid from the UNMODIFIED weights
            quantize_range_kernel(quant_t, g.scale[:, 0].contiguous(), q_t, g.qzero[:, 0].contiguous(), float(g.maxq),
                                  u, w_t, error_t, i1, i2)
        quant, qweight, error = quant_t.t(), q_t.t().to(torch.float32), error_t.t()
        debug = {"weight_t": w_t, "quant_t": quant_t, "error_t": error_t}
    else:  # gptq_py.py:158-193
        assert block_sizes is not None
        u = hook.hessian_inv
        quant = torch.empty_like(w)
        qweight = torch.empty_like(w)
        error = torch.empty_like(w)
        bbounds = [0] + torch.cumsum(block_sizes, 0).tolist()
        gi = 0
        g = None
        for i1, i2 in zip(bbounds[:-1], bbounds[1:]):
            blk = w[:, i1:i2].clone()
            eblk = torch.empty_like(blk)
            for j in range(i1, i2):
                if gi < len(bounds) - 1 and j == bounds[gi]:
                    # the grid sees w as it was when the block began; the search looks at the block
                    g = new_grid(w[:, j:j + int(group_sizes[gi])], int(group_bit_widths[gi]), w[:, i1:i2])
                    gi += 1
                col = blk[:, j - i1:j - i1 + 1]
                qc = quantize(col, g.scale, g.qzero, g.maxq)
                dq = dequantize(qc, g.scale, g.qzero)
                e = (col - dq) / u[j, j]
                qweight[:, j:j + 1] = qc
                quant[:, j:j + 1] = dq
                error[:, j:j + 1] = e
                eblk[:, j - i1:j - i1 + 1] = e
                blk[:, j - i1 + 1:] -= e @ u[j:j + 1, j + 1:i2]
            w[:, i2:] -= eblk @ u[i1:i2, i2:]

    metrics = {
        "gptq_error": float((error ** 2.0).mean()),
        "gptq_norm": float(((w_ref / hook.hessian_inv.diagonal()) ** 2.0).mean()) if hook is not None
        else float((error ** 2.0).mean()),
    }
    # gptq_py.py:202-229: with scale quantisation the groups are stored sorted by
    # (bit width, size, -index), descending; otherwise in place
    n_groups = len(group_sizes)
    order = list(range(n_groups))
    if scale_bit_width is not None:
        order.sort(key=lambda k: (int(group_bit_widths[k]), int(group_sizes[k]), -k), reverse=True)
    idx_c = torch.cat([torch.arange(bounds[k], bounds[k + 1]) for k in order])
    idx_g = torch.tensor(order, dtype=torch.int64)
    exl2 = grids[0].qscale is not None
    scale = torch.cat([g.scale for g in grids], dim=-1).to(dtype)[:, idx_g]
    qzero = torch.cat([g.qzero for g in grids], dim=-1).to(torch.uint8)[:, idx_g]
    qscale = torch.cat([g.qscale for g in grids], dim=-1).to(torch.uint8)[:, idx_g] if exl2 else None
    sscale = torch.cat([g.sscale for g in grids], dim=-1).to(dtype)[0, idx_g] if exl2 else None
    if hook is not None and hook.perm is not None:
        perm = hook.perm.to(torch.int16)[idx_c]
    else:
        perm = torch.arange(n_cols, dtype=torch.int16)[idx_c]
    perm_inv = torch.argsort(perm).to(torch.int16)
    quant_sorted = quant[:, idx_c]
    quant_unperm = quant_sorted[:, perm_inv.to(torch.int32)]
    metrics["l2.0_error"] = float(((quant_unperm.float() - w_ref.float()).abs() ** 2.0).mean())
    metrics["l2.0_norm"] = float((w_ref.abs() ** 2.0).mean())
    meta = {
        "qweight": qweight.to(torch.uint8)[:, idx_c].contiguous(),
        "scale": scale.contiguous(),
        "qzero": qzero.contiguous(),
        "qscale": qscale,
        "sscale": sscale,
        "perm_inv": perm_inv,
        "group_sizes": group_sizes.to(torch.int16)[idx_g],
        "group_bit_widths": group_bit_widths.to(torch.uint8)[idx_g],
        "scale_bit_width": None if scale_bit_width is None else torch.as_tensor(scale_bit_width).to(torch.uint8),
    }
    debug["quant"] = quant
    return {"quant_meta": meta, "metrics": metrics, "debug": debug}


# ---------------------------------------------------------------------------------------
# Reconstruction (quant.py:222-237, 262-279)
# ---------------------------------------------------------------------------------------

def construct_weight(meta: dict, dtype: torch.dtype = torch.float16, unpermute: bool = True) -> torch.Tensor:
    """(int16 q - int16 z) * scale evaluated in ``dtype`` per group, then columns un-permuted."""
    q = meta["qweight"].to(torch.int16)
    z = meta["qzero"].to(torch.int16)
    if meta.get("sscale") is not None:  # quant.py:266-269, 84-85: int16 level squared times the step in `dtype`
        qs = meta["qscale"].to(torch.int16)
        s = (qs * qs) * meta["sscale"].to(dtype).reshape(1, -1)
    else:
        s = meta["scale"].to(dtype)
    bounds = [0] + torch.cumsum(meta["group_sizes"].to(torch.int64), 0).tolist()
    w = torch.empty(q.shape, dtype=dtype)
    for k, (i1, i2) in enumerate(zip(bounds[:-1], bounds[1:])):
        w[:, i1:i2] = (q[:, i1:i2] - z[:, k:k + 1]) * s[:, k:k + 1]
    if unpermute and meta.get("perm_inv") is not None:
        w = w[:, meta["perm_inv"].to(torch.int64)]
    return w


# ---------------------------------------------------------------------------------------
# Packed layout (defined here; the reference stores one value per byte, gptq_py.py:217)
# ---------------------------------------------------------------------------------------
# A row of ``n`` b-bit values is a little-endian bit stream: value i occupies bits
# [i*b, i*b + b) of the stream, the stream is cut into uint32 words (word w holds stream
# bits [32w, 32w+32)).  For b = 4 that is 8 values per word, value i in bits 4*(i%8);
# for b = 3, 32 values fill exactly 3 words.  Rows are padded with zero bits to a whole word.

def pack_rows(values: torch.Tensor, bits: int) -> torch.Tensor:
    v = values.to(torch.int64).numpy()
    rows, n = v.shape
    assert v.min(initial=0) >= 0 and v.max(initial=0) < (1 << bits)
    n_words = (n * bits + 31) // 32
    out = np.zeros((rows, n_words), dtype=np.uint64)
    for i in range(n):
        bit = i * bits
        wi, sh = bit // 32, bit % 32
        out[:, wi] |= (v[:, i].astype(np.uint64) << np.uint64(sh)) & np.uint64(0xFFFFFFFF)
        if sh + bits > 32:
            out[:, wi + 1] |= v[:, i].astype(np.uint64) >> np.uint64(32 - sh)
    return torch.from_numpy(out.astype(np.uint32).view(np.int32))


def unpack_rows(packed: torch.Tensor, bits: int, n: int) -> torch.Tensor:
    p = packed.numpy().view(np.uint32).astype(np.uint64)
    rows = p.shape[0]
    out = np.zeros((rows, n), dtype=np.uint8)
    mask = np.uint64((1 << bits) - 1)
    for i in range(n):
        bit = i * bits
        wi, sh = bit // 32, bit % 32
        val = p[:, wi] >> np.uint64(sh)
        if sh + bits > 32:
            val |= p[:, wi + 1] << np.uint64(32 - sh)
        out[:, i] = (val & mask).astype(np.uint8)
    return torch.from_numpy(out)


def pack_rows_w3_kblock(values: torch.Tensor) -> torch.Tensor:
    """The GEMM operand layout of dense 3-bit levels (defined here, no reference counterpart):
    (rows, n) -> (n/64, rows, 6) words.  Every run of 32 levels of a row takes 3 words: level v < 30
    sits in word v // 10 at bit offset 3 * ((v % 10) // 2), in the low 16-bit half if v is even and
    the high half if odd; bit j of level 30 is bit 15 of word j, bit j of level 31 is bit 31."""
    v = values.to(torch.int64).numpy()
    rows, n = v.shape
    assert n % 64 == 0 and v.min(initial=0) >= 0 and v.max(initial=0) < 8
    out = np.zeros((n // 64, rows, 6), dtype=np.uint64)
    for i in range(n):
        kb, chunk, lv = i // 64, (i % 64) // 32, i % 32
        col = v[:, i].astype(np.uint64)
        if lv < 30:
            word, sh = lv // 10, 3 * ((lv % 10) // 2) + 16 * (lv & 1)
            out[kb, :, chunk * 3 + word] |= col << np.uint64(sh)
        else:
            sh = 15 if lv == 30 else 31
            for j in range(3):
                out[kb, :, chunk * 3 + j] |= ((col >> np.uint64(j)) & np.uint64(1)) << np.uint64(sh)
    return torch.from_numpy(out.astype(np.uint32).view(np.int32))


def w4a16_linear(x: torch.Tensor, meta: dict) -> torch.Tensor:
    """What QuantLinear.forward must approximate: the reference's reconstruct-then-dense
    path (llama.py:172-177) with the fp16 weight and an exact (fp64) contraction."""
    w = construct_weight(meta, torch.float16, unpermute=True)
    return (x.to(torch.float64) @ w.to(torch.float64).t()).to(torch.float32)
