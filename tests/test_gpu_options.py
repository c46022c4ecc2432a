"""gptq_quant's remaining knobs on the GPU, through the C ABI: the shrink search (quant_mse), the
EXL2 scale quantisation (scale_bit_width) with the sorted storage of mixed groups, and PyTorch-loop
blocks that differ from the groups.

Two references are used.  The fixtures f8_* are outputs of the reference itself.  Where the shrink
search is on, the reference decides between candidates by fp32 sums of fp32 powers, the HIP kernel
by float64 sums; the two agree except where two candidates' errors are within fp32 rounding of each
other, so against the fixtures a bounded fraction of grids may sit one candidate apart, and the
bit-exact comparison is against the CPU oracle evaluated with float64 sums (mse_exact=True), whose
fp32-sum twin is pinned to the fixtures bit for bit in test_oracle_golden.py.
"""
import numpy as np
import pytest
import torch

from tests.conftest import load_golden, t

pytestmark = pytest.mark.gpu


def _hooks(amd, oracle, g, gpu):
    """GPU hook and oracle hook carrying the reference's U and perm."""
    if "hessian_inv" not in g:
        return None, None
    hook = amd.HessianHook()
    hook.hessian_inv = t(g["hessian_inv"]).to(gpu)
    hook.hessian = torch.ones(1, device=gpu)
    oh = oracle.OracleHessian()
    oh.hessian_inv = t(g["hessian_inv"])
    oh.hessian = torch.ones(g["hessian_inv"].shape[0], g["hessian_inv"].shape[0])
    if bool(g["act_order"]):
        hook.perm = t(g["perm"]).to(gpu)
        hook.perm_inv = torch.argsort(hook.perm)
        oh.perm = t(g["perm"])
        oh.perm_inv = torch.argsort(oh.perm)
    return hook, oh


def _run(amd, oracle, name, gpu):
    g = load_golden(name)
    hook, oh = _hooks(amd, oracle, g, gpu)
    gs, gb, bs = t(g["group_sizes"]), t(g["group_bit_widths"]), t(g["block_sizes"])
    sbw = torch.tensor(int(g["scale_bits"])) if "scale_bits" in g else None
    kernel = "kernelsem" in name
    res = amd.gptq_quant(t(g["w"]).to(gpu), hook, gs, gb, scale_bit_width=sbw, gptq_use_kernel=kernel,
                         gptq_block_sizes=bs, quant_symmetric=bool(g["sym"]), quant_mse=bool(g["mse"]))
    want = oracle.gptq_quant(t(g["w"]), oh, gs, gb, use_kernel_semantics=kernel, block_sizes=bs, symmetric=bool(g["sym"]),
                             scale_bit_width=None if sbw is None else int(sbw), mse=bool(g["mse"]), mse_exact=True)
    return g, res, want


META_KEYS = ("qweight", "scale", "qzero", "qscale", "sscale", "perm_inv", "group_sizes", "group_bit_widths", "scale_bit_width")


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a.dtype == b.dtype and torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize("name", ["f8_kernelsem_mse_64x256_b4", "f8_kernelsem_exl2_64x256_mixed",
                                  "f8_kernelsem_exl2_nomse_64x256", "f8_rtn_mse_exl2_64x256"])
def test_static_grids_with_options_bit_exact(amd, oracle, gpu, name):
    """Native-loop semantics and round-to-nearest: every output identical to the oracle with
    float64 error sums; against the reference's own output, identical where no search is involved,
    else at least 90 % of the grids identical and the rest one shrink candidate apart."""
    g, res, want = _run(amd, oracle, name, gpu)
    meta = res["quant_meta"]
    for k in META_KEYS:
        assert _same(meta[k], want["quant_meta"][k]), k
    for k in ("gptq_error", "gptq_norm", "l2.0_error", "l2.0_norm"):
        assert res["metrics"][k] == pytest.approx(want["metrics"][k], rel=1e-4, abs=1e-30), k
    assert torch.equal(meta["group_sizes"], t(g["meta_group_sizes"]))
    assert torch.equal(meta["group_bit_widths"], t(g["meta_group_bit_widths"]))
    assert torch.equal(meta["perm_inv"], t(g["perm_inv"]))
    assert torch.equal(meta["qzero"], t(g["qzero"]))
    if not bool(g["mse"]):
        for k in ("qweight", "scale", "qscale", "sscale"):
            assert torch.equal(meta[k], t(g[k])), k
    else:
        s, sr = meta["scale"].float(), t(g["scale"]).float()
        assert (s == sr).float().mean() >= 0.9
        assert ((s - sr).abs() <= 0.025 * sr).all()   # candidates are 1 % apart; the EXL2 grid adds its own step
        if "sscale" in g:
            assert ((meta["sscale"].float() - t(g["sscale"]).float()).abs() <= 0.011 * t(g["sscale"]).float()).all()
            assert (meta["qscale"].int() - t(g["qscale"]).int()).abs().max() <= 1


# levels that may differ from the reference's: the count measured on an MI355X (none), as in test_gpu_parity.py
PYSEM_LEVELS_OFF = {"f8_pysem_blocks128_g64_64x256": 0, "f8_pysem_blocks64_g128_64x256": 0}


@pytest.mark.parametrize("name,exact_groups", [("f8_pysem_blocks128_g64_64x256", 2), ("f8_pysem_blocks64_g128_64x256", 1)])
def test_python_semantics_blocks_differ_from_groups(amd, oracle, gpu, name, exact_groups):
    """The PyTorch loop with blocks of 128 over groups of 64 (a group that starts inside a block takes
    its grid from the weights as they were when the block began: the first TWO grids come from the
    unmodified weights and are exact) and with blocks of 64 over groups of 128.  Later columns
    depend on the summation order of the reference's fp32 matmuls: a level may differ by one; how many
    do is printed and held to the measured count."""
    g, res, _ = _run(amd, oracle, name, gpu)
    meta = res["quant_meta"]
    assert torch.equal(meta["scale"][:, :exact_groups], t(g["scale"])[:, :exact_groups])
    assert torch.equal(meta["qzero"][:, :exact_groups], t(g["qzero"])[:, :exact_groups])
    q, qr = meta["qweight"].int(), t(g["qweight"]).int()
    assert torch.equal(q[:, 0], qr[:, 0])
    assert (q - qr).abs().max() <= 1
    n_off = int((q != qr).sum())
    print(f"{name}: {n_off} of {q.numel()} levels differ from the reference's")
    assert n_off <= PYSEM_LEVELS_OFF[name]
    s, sr = meta["scale"].float(), t(g["scale"]).float()
    assert ((s - sr).abs() <= 2 * 2 ** -10 * sr.abs()).all()
    got = np.array([res["metrics"][k] for k in ("gptq_error", "gptq_norm", "l2.0_error", "l2.0_norm")])
    np.testing.assert_allclose(got, g["metrics"], rtol=2e-3)


def test_python_semantics_stale_grid_differs_from_a_fresh_one(amd, oracle, gpu):
    """The same layer with blocks == groups gives the second group a grid from UPDATED weights: the
    block structure is really honoured, not ignored."""
    g = load_golden("f8_pysem_blocks128_g64_64x256")
    hook, _ = _hooks(amd, oracle, g, gpu)
    gs, gb = t(g["group_sizes"]), t(g["group_bit_widths"])
    res = amd.gptq_quant(t(g["w"]).to(gpu), hook, gs, gb, gptq_use_kernel=False, gptq_block_sizes=gs, quant_mse=False)
    assert torch.equal(res["quant_meta"]["scale"][:, 0], t(g["scale"])[:, 0])
    assert not torch.equal(res["quant_meta"]["scale"][:, 1], t(g["scale"])[:, 1])


def test_python_semantics_with_search_and_scale_quantisation(amd, oracle, gpu):
    """Everything at once in the PyTorch loop (search over the BLOCK's columns, EXL2 scales).  The
    first block's two groups see unmodified weights: exact against the float64-sum oracle.  Against
    the reference's output: levels of the quantised scales at most one apart, 90 % identical."""
    g, res, want = _run(amd, oracle, "f8_pysem_mse_exl2_64x256", gpu)
    meta, wm = res["quant_meta"], want["quant_meta"]
    for k in ("scale", "qzero", "qscale"):
        assert torch.equal(meta[k][:, :2], wm[k][:, :2]), k
    assert torch.equal(meta["sscale"][:2], wm["sscale"][:2])
    assert torch.equal(meta["qweight"][:, :1], wm["qweight"][:, :1])
    dq = (meta["qscale"].int() - t(g["qscale"]).int()).abs()
    assert dq.max() <= 1 and (dq == 0).float().mean() >= 0.9
    assert ((meta["sscale"].float() - t(g["sscale"]).float()).abs() <= 0.011 * t(g["sscale"]).float()).all()
    assert torch.equal(meta["perm_inv"], t(g["perm_inv"]))
    assert int(meta["scale_bit_width"]) == 4


@pytest.mark.parametrize("name", ["f8_kernelsem_exl2_64x256_mixed", "f8_kernelsem_exl2_nomse_64x256",
                                  "f8_pysem_mse_exl2_64x256", "f8_rtn_mse_exl2_64x256"])
def test_reconstruct_from_reference_exl2_meta(amd, gpu, name):
    """reconstruct_nn_linear on the REFERENCE's EXL2 quant_meta (qscale^2 * sscale scales, sorted
    mixed groups): the dense fp16 weight bit for bit."""
    g = load_golden(name)
    meta = {"qweight": t(g["qweight"]), "scale": t(g["scale"]), "qzero": t(g["qzero"]), "qscale": t(g["qscale"]),
            "sscale": t(g["sscale"]), "perm_inv": t(g["perm_inv"]), "group_sizes": t(g["meta_group_sizes"]),
            "group_bit_widths": t(g["meta_group_bit_widths"])}
    lin = amd.reconstruct_nn_linear(meta, dtype=torch.float16, device=gpu)
    assert torch.equal(lin.weight.data.cpu(), t(g["recon_weight"]))
    y = lin(t(g["lin_x"]).to(gpu)).cpu()
    assert torch.allclose(y.float(), t(g["lin_y"]).float(), rtol=2e-3, atol=2e-3)


def test_quantlinear_from_exl2_meta(amd, gpu):
    """The packed W4A16 path takes the same scales the dense reconstruction uses."""
    g = load_golden("f8_kernelsem_exl2_nomse_64x256")
    meta = {"qweight": t(g["qweight"]), "scale": t(g["scale"]), "qzero": t(g["qzero"]), "qscale": t(g["qscale"]),
            "sscale": t(g["sscale"]), "perm_inv": t(g["perm_inv"]), "group_sizes": t(g["meta_group_sizes"]),
            "group_bit_widths": t(g["meta_group_bit_widths"])}
    layer = amd.QuantLinear.from_quant_meta(meta, gpu)
    x = torch.randn(96, 256, generator=torch.Generator().manual_seed(5)).half().to(gpu)
    want = x.double() @ t(g["recon_weight"]).to(gpu).double().t()
    got = layer(x).double()
    assert (got - want).abs().max() <= 2e-3 * want.abs().max() + 1e-3


def test_search_and_scale_quantisation_llama_columns(amd, oracle, gpu):
    """A 256-row slice of a 4096-column layer, 4-bit g128, search and 4-bit scales on, native-loop
    semantics: the GPU's U and perm go into the oracle, every output identical."""
    c, r = 4096, 256
    gen = torch.Generator().manual_seed(77)
    x = (torch.randn(2, 4096, c, generator=gen) * (1 + 3 * torch.arange(c) / c)).half().to(gpu)
    w = (torch.randn(r, c, generator=gen) * 0.02).half().to(gpu)
    hook = amd.HessianHook()
    hook.add_batch(x)
    hook.invert()
    gs = torch.full([c // 128], 128, dtype=torch.int32)
    gb = torch.full([c // 128], 4, dtype=torch.int32)
    res = amd.gptq_quant(w, hook, gs, gb, scale_bit_width=torch.tensor(4), gptq_use_kernel=True, gptq_block_sizes=gs,
                         quant_mse=True)
    oh = oracle.OracleHessian()
    oh.hessian, oh.hessian_inv = hook.hessian.cpu(), hook.hessian_inv.cpu()
    oh.perm, oh.perm_inv = hook.perm.cpu(), hook.perm_inv.cpu()
    want = oracle.gptq_quant(w.cpu(), oh, gs, gb, use_kernel_semantics=True, scale_bit_width=4, mse=True, mse_exact=True)
    for k in META_KEYS:
        assert _same(res["quant_meta"][k], want["quant_meta"][k]), k
    # the search can only lower the round-to-nearest error of the grid it starts from
    plain = amd.gptq_quant(w, hook, gs, gb, scale_bit_width=torch.tensor(4), gptq_use_kernel=True, gptq_block_sizes=gs,
                           quant_mse=False)
    assert res["metrics"]["l2.0_error"] < plain["metrics"]["l2.0_error"] * 1.05


def test_option_errors(amd, gpu):
    g = load_golden("f7_rtn_64x256_b4")
    w = t(g["w"]).to(gpu)
    gs, gb = t(g["group_sizes"]), t(g["group_bit_widths"])
    with pytest.raises(NotImplementedError):   # the exllamav2 kernel branch of Quantizer.mse
        amd.gptq_quant(w, None, gs, gb, quant_mse=True, quant_use_kernel=True)
    with pytest.raises(ValueError):            # more shrink candidates than the kernel takes
        amd.gptq_quant(w, None, gs, gb, quant_mse=True, quant_n_grid=1000)
    with pytest.raises(ValueError):
        amd.gptq_quant(w, None, gs, gb, scale_bit_width=torch.tensor(9), quant_mse=False)
    hook = amd.HessianHook()
    hook.hessian_inv = torch.eye(256, device=gpu)
    hook.hessian = torch.ones(1, device=gpu)
    with pytest.raises(ValueError):            # blocks must cover the columns
        amd.gptq_quant(w, hook, gs, gb, gptq_use_kernel=False, gptq_block_sizes=torch.tensor([128, 64]), quant_mse=False)


@pytest.mark.parametrize("m,r,c,gsz", [(300, 200, 256, 64), (512, 384, 512, 128), (2048, 4096, 4096, 128),
                                       (4200, 4352, 512, 64), (4352, 4100, 1024, 128)])
def test_w4a16_gemm_bf16(amd, gpu, m, r, c, gsz):
    """bf16 activations / scales / output: the product equals x @ W^T for W = bf16((q - z) * scale)
    (one rounding, what reconstruct_nn_linear builds for a bf16 model) to bf16 output rounding."""
    gen = torch.Generator().manual_seed(m + r + c)
    q = torch.randint(0, 16, (r, c), generator=gen, dtype=torch.uint8).to(gpu)
    scale = (torch.rand(r, c // gsz, generator=gen) * 0.01 + 0.002).to(torch.bfloat16).to(gpu)
    zero = torch.randint(0, 16, (r, c // gsz), generator=gen, dtype=torch.uint8).to(gpu)
    x = torch.randn(m, c, generator=gen).to(torch.bfloat16).to(gpu)
    meta = {"qweight": q, "scale": scale, "qzero": zero, "qscale": None, "sscale": None, "perm_inv": None,
            "group_sizes": torch.full([c // gsz], gsz, dtype=torch.int16), "group_bit_widths": torch.full([c // gsz], 4, dtype=torch.uint8)}
    w = amd.reconstruct_nn_linear(meta, dtype=torch.bfloat16, device=gpu).weight.data
    # the dense weight itself: (int q - int z) * scale evaluated in bf16
    want_w = ((q.int() - zero.int().repeat_interleave(gsz, dim=1)).to(torch.bfloat16) * scale.repeat_interleave(gsz, dim=1))
    assert torch.equal(w, want_w)
    layer = amd.QuantLinear.from_quant_meta(meta, gpu)
    assert layer.dtype == torch.bfloat16 and layer.scales.dtype == torch.bfloat16
    y = layer(x)
    assert y.dtype == torch.bfloat16 and y.shape == (m, r)
    ref = x.double() @ w.double().t()
    err = (y.double() - ref).abs().max()
    assert err <= 2 ** -7 * ref.abs().max() + 1e-3, (float(err), float(ref.abs().max()))
    with pytest.raises(NotImplementedError):
        layer(x.half())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,r,gsz", [(512, 320, 64), (4352, 4352, 128)])
def test_w4a16_gemm_identity_is_dequantised_weight(amd, gpu, dtype, n, r, gsz):
    """x = identity: every output is one dequantised weight passed through the accumulator unchanged,
    so the product equals the dense reconstruct_nn_linear weight bit for bit -- the in-kernel
    expansion of the nibbles rounds as the reference's (q - z) * scale in the tensor dtype does.
    The larger case runs on the 256-tile kernel, the smaller on the 128-tile one."""
    gen = torch.Generator().manual_seed(n + r)
    q = torch.randint(0, 16, (r, n), generator=gen, dtype=torch.uint8).to(gpu)
    scale = (torch.rand(r, n // gsz, generator=gen) * 0.01 + 0.002).to(dtype).to(gpu)
    zero = torch.randint(0, 16, (r, n // gsz), generator=gen, dtype=torch.uint8).to(gpu)
    meta = {"qweight": q, "scale": scale, "qzero": zero, "qscale": None, "sscale": None, "perm_inv": None,
            "group_sizes": torch.full([n // gsz], gsz, dtype=torch.int16), "group_bit_widths": torch.full([n // gsz], 4, dtype=torch.uint8)}
    w = amd.reconstruct_nn_linear(meta, dtype=dtype, device=gpu).weight.data
    layer = amd.QuantLinear.from_quant_meta(meta, gpu)
    y = layer(torch.eye(n, dtype=dtype, device=gpu))
    assert torch.equal(y, w.t())


@pytest.mark.parametrize("bits", [2, 3])
def test_quantlinear_narrow_bit_widths(amd, gpu, bits):
    """3- and 2-bit layers (the Llama-2-70B configuration quantises to 3 bits) run through the same
    packed GEMM in 4-bit containers: identity input returns the dense reconstruct_nn_linear weight
    bit for bit, unpack() returns quant_meta unchanged, values beyond the width are refused."""
    n, r, gsz = 512, 320, 128
    gen = torch.Generator().manual_seed(bits)
    q = torch.randint(0, 1 << bits, (r, n), generator=gen, dtype=torch.uint8).to(gpu)
    scale = (torch.rand(r, n // gsz, generator=gen) * 0.01 + 0.002).half().to(gpu)
    zero = torch.randint(0, 1 << bits, (r, n // gsz), generator=gen, dtype=torch.uint8).to(gpu)
    perm_inv = torch.randperm(n, generator=gen).to(torch.int16)
    meta = {"qweight": q, "scale": scale, "qzero": zero, "qscale": None, "sscale": None, "perm_inv": perm_inv,
            "group_sizes": torch.full([n // gsz], gsz, dtype=torch.int16), "group_bit_widths": torch.full([n // gsz], bits, dtype=torch.uint8)}
    w = amd.reconstruct_nn_linear(meta, dtype=torch.float16, device=gpu).weight.data
    layer = amd.QuantLinear.from_quant_meta(meta, gpu)
    assert layer.bits == bits
    assert torch.equal(layer(torch.eye(n, dtype=torch.float16, device=gpu)), w.t())
    back = layer.unpack()
    assert torch.equal(back["qweight"], q) and torch.equal(back["qzero"], zero) and torch.equal(back["scale"], scale)
    assert torch.equal(back["perm_inv"].cpu(), perm_inv) and int(back["group_bit_widths"][0]) == bits
    bad = dict(meta, qweight=q.clone())
    bad["qweight"][0, 0] = 1 << bits
    with pytest.raises(ValueError):
        amd.QuantLinear.from_quant_meta(bad, gpu)
