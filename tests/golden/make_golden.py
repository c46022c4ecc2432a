#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING the reference (imported from /root/reference).

Run once in the build container:  python tests/golden/make_golden.py
The fixtures (inputs + the reference's outputs) are committed; the reference itself never
travels.  Nothing here runs at test time.

How the reference is imported: ``gptq_py.py`` does ``from gptq import accumulate_hessian,
gptq_quantize_range`` (gptq_py.py:3) where ``gptq`` is the CUDA extension package that cannot
be built here (needs nvcc + CUTLASS).  A stand-in module named ``gptq`` is placed in
``sys.modules`` for the import to succeed:
  * ``accumulate_hessian`` raises -- the fixtures only use ``add_batch(use_kernel=False)``;
  * ``gptq_quantize_range`` is our C restatement of gptq_cpp.cpp/gptq_cpp_kernel.cu
    (oracle/kernel_path_ref.c).  Fixtures named ``*_kernelsem_*`` therefore pin the
    reference's Python glue (prologue, grid from unmodified weights, collation, metrics)
    around OUR restatement of the native loop, not the CUDA code itself.
Fixtures named ``*_pysem_*`` are produced by reference code only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import gptq_oracle as oracle  # noqa: E402

stub = types.ModuleType("gptq")


def _no_native(*_a, **_k):
    raise RuntimeError("native accumulate_hessian is not available in the fixture generator")


def _native_quantize_range(quant, scale, out_q, qzero, maxq, hessian_inv, weights, error, a, b):
    oracle.quantize_range_kernel(quant, scale, out_q, qzero, maxq, hessian_inv, weights, error, a, b)


stub.accumulate_hessian = _no_native
stub.gptq_quantize_range = _native_quantize_range
sys.modules["gptq"] = stub
sys.path.insert(0, REF)
import gptq_py as ref_gptq  # noqa: E402  (the reference)
import quant as ref_quant  # noqa: E402  (the reference)

torch.cuda.set_device = lambda *_a, **_k: None  # gptq_py.py:127 calls it on a CPU tensor


def np_of(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        if t.dtype == torch.bfloat16:
            return t.view(torch.int16).numpy().view(np.uint16)
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrays):
    arrays = {k: np_of(v) for k, v in arrays.items() if v is not None}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  {sorted(arrays)}")


def calib(gen, n_seq, n_tok, c, dtype=torch.float16):
    # non-uniform column energy so that act-order is a real permutation
    x = torch.randn(n_seq, n_tok, c, generator=gen) * (1.0 + 3.0 * torch.arange(c) / c)
    return x.to(dtype)


def make_hook(xs, dead_col=None):
    hook = ref_gptq.HessianHook()
    for x in xs:
        if dead_col is not None:
            x = x.clone()
            x[..., dead_col] = 0
        hook.add_batch(x, use_kernel=False)
    return hook


def f1_xtx():
    for dtype, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        gen = torch.Generator().manual_seed(101)
        batches = [calib(gen, 2, 64, 256, dtype), calib(gen, 1, 128, 256, dtype),
                   calib(gen, 3, 32, 256, dtype), calib(gen, 1, 160, 256, dtype)[0]]  # last one 2-D
        hook = ref_gptq.HessianHook()
        out = {}
        for i, x in enumerate(batches):
            hook.add_batch(x, use_kernel=False)
            out[f"x{i}"] = x
            out[f"h{i}"] = hook.hessian.clone()
            out[f"n{i}"] = np.int64(hook.n_samples)
        save(f"f1_xtx_{tag}", **out)


def f2_invert():
    gen = torch.Generator().manual_seed(202)
    c = 256
    xs = [calib(gen, 4, 96, c)]
    cases = {
        "act": dict(act_order=True, dead=None, xs=xs),
        "noact": dict(act_order=False, dead=None, xs=xs),
        "dead": dict(act_order=True, dead=17, xs=xs),
        # rank-deficient Hessian (64 tokens for 256 columns): only the damping makes it positive
        # definite.  (A failing factorisation cannot be pinned: LAPACK on CPU returns garbage
        # without NaN, so the reference's NaN-triggered retry never fires here.)
        "lowrank": dict(act_order=True, dead=None, xs=[calib(gen, 1, 64, c)]),
    }
    for name, cfg in cases.items():
        hook = make_hook(cfg["xs"], cfg["dead"])
        h_raw = hook.hessian.clone()
        n = hook.n_samples
        damp_ratio = cfg.get("damp", 1e-2)
        # count retries the way the reference loop does, by watching the diagonal grow
        u = hook.invert(damp_ratio=damp_ratio, act_order=cfg["act_order"])
        hs = h_raw * (2.0 / n)
        d = hs.diagonal().clone()
        d[d == 0] = 1.0
        damp = damp_ratio * d.mean()
        perm = hook.perm if hook.perm is not None else torch.arange(c)
        tries = int(torch.round((hook.hessian.diagonal()[0] - d[perm][0]) / damp).item())
        assert tries == 1 and not torch.isnan(u).any() and (u.diagonal() > 0).all()
        save(f"f2_invert_{name}", h_raw=h_raw, n_samples=np.int64(n), damp_ratio=np.float64(damp_ratio),
             act_order=np.bool_(cfg["act_order"]), hessian_after=hook.hessian, perm=hook.perm,
             perm_inv=hook.perm_inv, hessian_inv=u, tries=np.int64(tries))


def f3_find_params():
    gen = torch.Generator().manual_seed(303)
    w = torch.randn(64, 128, generator=gen) * 0.02
    w[0] = w[0].abs() + 0.01      # all positive
    w[1] = -w[1].abs() - 0.01     # all negative
    w[2] = 0.0                    # all zero
    w[3, :] = 0.5                 # constant
    out = {"w": w}
    for bits in (2, 3, 4, 8):
        for sym in (False, True):
            q = ref_quant.Quantizer()
            q.find_params(w, bit_width=torch.tensor(bits), sym=sym)
            tag = f"b{bits}_{'sym' if sym else 'asym'}"
            out[f"scale_{tag}"] = q.scale
            out[f"qzero_{tag}"] = q.qzero
            out[f"maxq_{tag}"] = np.float64(float(q.maxq))
            out[f"q_{tag}"] = q.quantize(w)
            out[f"dq_{tag}"] = q.dequantize(q.quantize(w))
    save("f3_find_params", **out)


def quant_case(tag, r, c, bits, seed, use_kernel, act_order=True, with_hook=True, group_sizes=None, group_bits=None,
               block_sizes=None, scale_bits=None, mse=False, sym=False):
    gen = torch.Generator().manual_seed(seed)
    w = (torch.randn(r, c, generator=gen) * 0.02).to(torch.float16)
    x = calib(gen, 4, max(96, c // 2), c)
    gs = torch.tensor(group_sizes, dtype=torch.int32) if group_sizes else torch.full([c // 128], 128, dtype=torch.int32)
    gb = torch.tensor(group_bits, dtype=torch.int32) if group_bits else torch.full([len(gs)], bits, dtype=torch.int32)
    bs = torch.tensor(block_sizes, dtype=torch.int32) if block_sizes else gs
    sbw = None if scale_bits is None else torch.tensor(scale_bits, dtype=torch.int32)
    hook = None
    extra = {}
    if with_hook:
        hook = make_hook([x])
        h_raw = hook.hessian.clone()
        hook.invert(damp_ratio=1e-2, act_order=act_order)
        extra = dict(h_raw=h_raw, n_samples=np.int64(hook.n_samples), hessian_inv=hook.hessian_inv,
                     perm=hook.perm)
    res = ref_gptq.gptq_quant(weight=w, hessian_hook=hook, group_sizes=gs, group_bit_widths=gb,
                              scale_bit_width=sbw, gptq_use_kernel=use_kernel, gptq_block_sizes=bs,
                              quant_symmetric=sym, quant_mse=mse, save_device=torch.device("cpu"))
    meta, metrics = res["quant_meta"], res["metrics"]
    lin = ref_quant.reconstruct_nn_linear(meta, dtype=torch.float16)
    xin = (torch.randn(8, c, generator=gen)).to(torch.float16)
    y = lin(xin)
    save(tag, w=w, group_sizes=gs, group_bit_widths=gb, block_sizes=bs, act_order=np.bool_(act_order),
         sym=np.bool_(sym), mse=np.bool_(mse), scale_bits=None if scale_bits is None else np.int64(scale_bits),
         qweight=meta["qweight"], scale=meta["scale"], qzero=meta["qzero"], perm_inv=meta["perm_inv"],
         qscale=meta["qscale"], sscale=meta["sscale"],
         meta_group_sizes=meta["group_sizes"], meta_group_bit_widths=meta["group_bit_widths"],
         metrics=np.array([metrics["gptq_error"], metrics["gptq_norm"], metrics["l2.0_error"],
                           metrics["l2.0_norm"]], dtype=np.float64),
         recon_weight=lin.weight.data, lin_x=xin, lin_y=y, **extra)


def f9_grid_options():
    """Quantizer.find_params with scale quantisation and Quantizer.mse on their own."""
    gen = torch.Generator().manual_seed(909)
    w = torch.randn(48, 96, generator=gen) * 0.02
    w[0] = 0.0
    w[1, :] = 0.25
    out = {"w": w}
    for bits, sbw, mse in ((4, 4, False), (3, 4, True), (4, None, True), (4, 6, True)):
        q = ref_quant.Quantizer()
        q.find_params(w, bit_width=torch.tensor(bits), sym=False,
                      scale_bit_width=None if sbw is None else torch.tensor(sbw))
        if mse:
            q.mse(w)
        tag = f"b{bits}_s{sbw}_{'mse' if mse else 'nomse'}"
        out[f"scale_{tag}"] = q.scale
        out[f"qzero_{tag}"] = q.qzero
        if sbw is not None:
            out[f"qscale_{tag}"] = q.qscale
            out[f"sscale_{tag}"] = q.sscale
    save("f9_grid_options", **out)


def main():
    torch.manual_seed(0)
    f1_xtx()
    f2_invert()
    f3_find_params()
    quant_case("f4_pysem_64x256_b4", 64, 256, 4, 404, use_kernel=False)
    quant_case("f4_pysem_128x384_b3", 128, 384, 3, 405, use_kernel=False)
    quant_case("f4_pysem_64x256_b4_noact", 64, 256, 4, 406, use_kernel=False, act_order=False)
    quant_case("f5_kernelsem_64x256_b4", 64, 256, 4, 404, use_kernel=True)
    quant_case("f5_kernelsem_128x384_b3", 128, 384, 3, 405, use_kernel=True)
    quant_case("f5_kernelsem_64x256_b4_noact", 64, 256, 4, 406, use_kernel=True, act_order=False)
    quant_case("f7_rtn_64x256_b4", 64, 256, 4, 407, use_kernel=False, with_hook=False)
    # the knobs the reference's driver leaves off: shrink search, scale quantisation with mixed
    # groups (their stored order is sorted), PyTorch-loop blocks that differ from the groups
    quant_case("f8_kernelsem_mse_64x256_b4", 64, 256, 4, 408, use_kernel=True, mse=True)
    quant_case("f8_kernelsem_exl2_64x256_mixed", 64, 256, 4, 409, use_kernel=True, mse=True, scale_bits=4,
               group_sizes=[64, 128, 32, 32], group_bits=[3, 4, 4, 3])
    quant_case("f8_kernelsem_exl2_nomse_64x256", 64, 256, 4, 410, use_kernel=True, scale_bits=4)
    quant_case("f8_pysem_blocks128_g64_64x256", 64, 256, 4, 411, use_kernel=False, group_sizes=[64] * 4,
               block_sizes=[128, 128])
    quant_case("f8_pysem_blocks64_g128_64x256", 64, 256, 4, 412, use_kernel=False, block_sizes=[64] * 4)
    quant_case("f8_pysem_mse_exl2_64x256", 64, 256, 4, 413, use_kernel=False, mse=True, scale_bits=4,
               group_sizes=[64] * 4, block_sizes=[128, 128])
    quant_case("f8_rtn_mse_exl2_64x256", 64, 256, 4, 414, use_kernel=False, with_hook=False, mse=True, scale_bits=4,
               group_sizes=[128, 128], group_bits=[3, 4])
    f9_grid_options()


if __name__ == "__main__":
    main()
