"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GPTQ hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product package never does.

It restates, on CPU tensors, what the reference's Python layer computes
(citations are ``file:line`` in the reference checkout):

  =====================  ==========================================================
  here                   reference
  =====================  ==========================================================
  OracleHessian          gptq_py.py:9-66     HessianHook.add_batch(use_kernel=False), invert
  find_params            quant.py:89-118     Quantizer.find_params (min/max and symmetric grids)
  quantize / dequantize  quant.py:10-30
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


def find_params(x: torch.Tensor, bit_width, sym: bool = False):
    """x (..., R, C) -> scale, qzero of shape (..., R, 1), maxq scalar.  quant.py:102-118
    without the EXL2 scale-quantisation branch (switched off by llama.py:153)."""
    maxq = 2.0 ** bit_width - 1.0
    if sym:
        scale = x.abs().amax(dim=-1, keepdim=True) * (2.0 / maxq) + EPSILON
        qzero = torch.full_like(scale, float(torch.round(torch.as_tensor((maxq + 1.0) * 0.5))))
    else:
        hi = torch.relu(x.amax(dim=-1, keepdim=True))
        lo = -torch.relu(-x.amin(dim=-1, keepdim=True))
        scale = (hi - lo) / maxq + EPSILON
        qzero = torch.round(-lo / scale)
    return scale, qzero, maxq


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
               block_sizes: Optional[torch.Tensor] = None, symmetric: bool = False) -> dict:
    """Returns {'quant_meta': {...}, 'metrics': {...}, 'debug': {...}} with quant_meta and
    metrics as in gptq_py.py:240-251 (mse search and EXL2 scales are not restated: the
    driver switches both off, llama.py:153,157)."""
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
    scales, zeros = [], []
    debug: dict = {}

    if hook is None:  # gptq_py.py:101-124 round to nearest
        quant = torch.empty_like(w)
        qweight = torch.empty_like(w)
        for bw, i1, i2 in zip(group_bit_widths.tolist(), bounds[:-1], bounds[1:]):
            s, z, mq = find_params(w[:, i1:i2], bw, symmetric)
            scales.append(s)
            zeros.append(z)
            qweight[:, i1:i2] = quantize(w[:, i1:i2], s, z, mq)
            quant[:, i1:i2] = dequantize(qweight[:, i1:i2], s, z)
        error = torch.zeros((), dtype=torch.float32)
    elif use_kernel_semantics:  # gptq_py.py:125-157
        w_t = w.t().contiguous()
        quant_t = torch.empty_like(w_t)
        error_t = torch.empty_like(w_t)
        q_t = torch.empty_like(w_t, dtype=torch.int16)
        u = hook.hessian_inv.contiguous()
        for bw, i1, i2 in zip(group_bit_widths.tolist(), bounds[:-1], bounds[1:]):
            s, z, mq = find_params(w[:, i1:i2], bw, symmetric)  # grid from the UNMODIFIED weights
            scales.append(s)
            zeros.append(z)
            quantize_range_kernel(quant_t, s[:, 0].contiguous(), q_t, z[:, 0].contiguous(), float(mq),
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
        g = 0
        s = z = mq = None
        for i1, i2 in zip(bbounds[:-1], bbounds[1:]):
            blk = w[:, i1:i2].clone()
            eblk = torch.empty_like(blk)
            for j in range(i1, i2):
                if j == bounds[g]:
                    s, z, mq = find_params(w[:, j:j + int(group_sizes[g])], int(group_bit_widths[g]), symmetric)
                    scales.append(s)
                    zeros.append(z)
                    g += 1
                col = blk[:, j - i1:j - i1 + 1]
                qc = quantize(col, s, z, mq)
                dq = dequantize(qc, s, z)
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
    # gptq_py.py:216-229 with scale_bit_width None: groups keep their order
    scale = torch.cat(scales, dim=-1).to(dtype)
    qzero = torch.cat(zeros, dim=-1).to(torch.uint8)
    if hook is not None and hook.perm is not None:
        perm = hook.perm.to(torch.int16)
    else:
        perm = torch.arange(n_cols, dtype=torch.int16)
    perm_inv = torch.argsort(perm).to(torch.int16)
    quant_unperm = quant[:, perm_inv.to(torch.int32)]
    metrics["l2.0_error"] = float(((quant_unperm.float() - w_ref.float()).abs() ** 2.0).mean())
    metrics["l2.0_norm"] = float((w_ref.abs() ** 2.0).mean())
    meta = {
        "qweight": qweight.to(torch.uint8).contiguous(),
        "scale": scale.contiguous(),
        "qzero": qzero.contiguous(),
        "qscale": None,
        "sscale": None,
        "perm_inv": perm_inv,
        "group_sizes": group_sizes.to(torch.int16),
        "group_bit_widths": group_bit_widths.to(torch.uint8),
        "scale_bit_width": None,
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


def w4a16_linear(x: torch.Tensor, meta: dict) -> torch.Tensor:
    """What QuantLinear.forward must approximate: the reference's reconstruct-then-dense
    path (llama.py:172-177) with the fp16 weight and an exact (fp64) contraction."""
    w = construct_weight(meta, torch.float16, unpermute=True)
    return (x.to(torch.float64) @ w.to(torch.float64).t()).to(torch.float32)
