"""Surfaces and boundary behaviour that sit beside the main parity cases: the GPTQ
(add_batch / fasterquant) adapter, a Hessian that holds NaN, column ranges of the native op that
are not aligned to the blocked kernel, fp32 activations through add_batch(use_kernel=False), and
where reconstruct_nn_linear puts its result."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer_and_batches(gpu, r, c, n_batches, seed):
    gen = torch.Generator().manual_seed(seed)
    xs = [(torch.randn(2, 300, c, generator=gen) * (1 + 3 * torch.arange(c) / c)).half().to(gpu) for _ in range(n_batches)]
    lin = torch.nn.Linear(c, r, bias=False, dtype=torch.float16)
    lin.weight.data = (torch.randn(r, c, generator=gen) * 0.02).half()
    return lin.to(gpu), xs


@pytest.mark.parametrize("static_groups,blocksize", [(True, 128), (True, 64), (False, 128), (False, 256)])
def test_gptq_adapter_equals_hook_and_gptq_quant(amd, gpu, static_groups, blocksize):
    """GPTQ(layer).add_batch(...) / fasterquant(...) against HessianHook + gptq_quant on the same
    inputs: every tensor of quant_meta and every metric identical.  With static groups the block
    size does not enter the result; with dynamic groups it is the PyTorch loop's block."""
    r, c = 192, 512
    lin, xs = _layer_and_batches(gpu, r, c, 3, 21)
    g = amd.GPTQ(lin)
    hook = amd.HessianHook()
    for x in xs:
        g.add_batch(x, lin(x))
        hook.add_batch(x)
    assert torch.equal(g.hook.hessian, hook.hessian) and g.hook.n_samples == hook.n_samples == 6
    got = g.fasterquant(blocksize=blocksize, percdamp=0.01, groupsize=128, actorder=True, bits=4, sym=False,
                        static_groups=static_groups)
    hook.invert(damp_ratio=0.01, act_order=True)
    gs = torch.full([c // 128], 128, dtype=torch.int32)
    gb = torch.full([c // 128], 4, dtype=torch.int32)
    blocks = gs if static_groups else torch.full([c // blocksize], blocksize, dtype=torch.int32)
    want = amd.gptq_quant(lin.weight.data, hook, gs, gb, gptq_use_kernel=static_groups, gptq_block_sizes=blocks,
                          quant_mse=False, save_device=gpu)
    for k in ("qweight", "scale", "qzero", "perm_inv", "group_sizes", "group_bit_widths"):
        assert torch.equal(got["quant_meta"][k], want["quant_meta"][k]), k
    for k, v in want["metrics"].items():   # sums of doubles added by atomics: the order of the additions is not fixed
        assert got["metrics"][k] == pytest.approx(v, rel=1e-9), k
    if static_groups and blocksize != 128:
        # the same result as with the default block: the cut into blocks changes no bit
        g2 = amd.GPTQ(lin)
        for x in xs:
            g2.add_batch(x)
        ref = g2.fasterquant(blocksize=128, static_groups=True)
        assert torch.equal(ref["quant_meta"]["qweight"], got["quant_meta"]["qweight"])
    with pytest.raises(ValueError):
        amd.GPTQ(lin).fasterquant(blocksize=6)


def test_gptq_adapter_whole_row_groups(amd, oracle, gpu):
    """groupsize=-1 (one group per row) through the adapter against the CPU oracle given the GPU's U."""
    r, c = 64, 256
    lin, xs = _layer_and_batches(gpu, r, c, 2, 4)
    g = amd.GPTQ(lin)
    for x in xs:
        g.add_batch(x)
    got = g.fasterquant(groupsize=-1, actorder=False, bits=3)
    oh = oracle.OracleHessian()
    oh.hessian, oh.hessian_inv = g.hook.hessian.cpu(), g.hook.hessian_inv.cpu()
    gs, gb = torch.tensor([c], dtype=torch.int32), torch.tensor([3], dtype=torch.int32)
    want = oracle.gptq_quant(lin.weight.data.cpu(), oh, gs, gb, use_kernel_semantics=True)
    for k in ("qweight", "scale", "qzero", "perm_inv"):
        assert torch.equal(got["quant_meta"][k].cpu(), want["quant_meta"][k]), k


@pytest.mark.parametrize("where", ["diag", "offdiag_and_diag", "inf"])
def test_invert_with_nonfinite_hessian_fails_cleanly(amd, gpu, where):
    """fp16 activations that overflow put inf (and inf * 0 = NaN) into X^T X.  The ranking must stay a
    permutation (no out-of-bounds gather) and the call must fail with an error, not a GPU fault;
    the reference fails the same input with an assert after its retries (gptq_py.py:56-65)."""
    from gptq_v2_amd._lib import GqError
    c = 384
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(4 * c, c, generator=gen)
    h = (x.t() @ x).float()
    if where == "diag":
        h[7, 7] = float("nan")
        h[200, 200] = float("nan")
    elif where == "offdiag_and_diag":
        h[5, :] = float("nan")
        h[:, 5] = float("nan")
    else:
        h[9, 9] = float("inf")
    hook = amd.HessianHook()
    hook.hessian, hook.n_samples = h.to(gpu), 4
    with pytest.raises(GqError, match="not finite"):
        hook.invert()
    torch.cuda.synchronize()
    # the device is still healthy and a finite Hessian inverts afterwards
    hook2 = amd.HessianHook()
    hook2.hessian, hook2.n_samples = (x.t() @ x).float().to(gpu), 4
    u = hook2.invert()
    assert torch.isfinite(u).all() and hook2.tries == 1
    assert torch.equal(torch.sort(hook2.perm).values, torch.arange(c, device=gpu))


def _spd_factor(c, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(3 * c, c, generator=gen, dtype=torch.float64) * (1 + 3 * torch.arange(c) / c)
    h = x.t() @ x / c
    h += 0.01 * h.diagonal().mean() * torch.eye(c, dtype=torch.float64)
    return torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(h)), upper=True).float().contiguous()


@pytest.mark.parametrize("c,r,a,b", [(256, 64, 5, 131), (256, 64, 0, 3), (256, 68, 130, 131), (256, 64, 6, 7),
                                     (130, 66, 0, 130), (130, 66, 17, 94), (132, 67, 3, 129), (256, 64, 9, 9)])
def test_quantize_range_any_range(amd, oracle, gpu, c, r, a, b):
    """gptq.gptq_quantize_range takes any [a, b) and any R, C, as the reference does
    (gptq_cpp.cpp:51): heads and tails that are not multiples of 4, and shapes whose rows are not
    16-byte aligned, bit-identical to the C restatement."""
    u = _spd_factor(c, c + r)
    gen = torch.Generator().manual_seed(c * 7 + r + a)
    w_t = (torch.randn(c, r, generator=gen) * 0.02).contiguous()
    scale, zero, maxq = oracle.find_params(w_t[a:max(b, a + 1)].t(), 4)
    scale, zero = scale[:, 0].contiguous(), zero[:, 0].contiguous()
    want = dict(quant=torch.zeros(c, r), q=torch.zeros(c, r, dtype=torch.int16), w=w_t.clone(), err=torch.zeros(c, r))
    oracle.quantize_range_kernel(want["quant"], scale, want["q"], zero, maxq, u, want["w"], want["err"], a, b)
    got = dict(quant=torch.zeros(c, r, device=gpu), q=torch.zeros(c, r, dtype=torch.int16, device=gpu),
               w=w_t.clone().to(gpu), err=torch.zeros(c, r, device=gpu))
    amd.gptq_quantize_range(got["quant"], scale.to(gpu), got["q"], zero.to(gpu), maxq, u.to(gpu), got["w"], got["err"], a, b)
    for k in ("q", "quant", "err", "w"):
        assert torch.equal(got[k].cpu(), want[k]), k


@pytest.mark.parametrize("shape", [(2, 300, 256), (5, 64, 132), (130, 64)])
def test_add_batch_fp32_activations(amd, oracle, gpu, shape):
    """add_batch(use_kernel=False) casts to fp32 and accumulates fp32 products (gptq_py.py:31-33):
    fp32 inputs are accepted there, and only there; H is exactly symmetric and equals the oracle's
    fp32 addmm to summation accuracy.  Token counts that are not multiples of 4 included."""
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen)
    hook, oh = amd.HessianHook(), oracle.OracleHessian()
    for _ in range(2):
        hook.add_batch(x.to(gpu), use_kernel=False)
        oh.add_batch(x)
    assert hook.n_samples == oh.n_samples
    h = hook.hessian.cpu()
    assert torch.equal(h, h.t())
    want = 2 * (x.reshape(-1, shape[-1]).double().t() @ x.reshape(-1, shape[-1]).double())
    assert (h.double() - want).abs().max() <= 2e-6 * want.abs().max()
    assert (h - oh.hessian).abs().max() <= 4e-6 * oh.hessian.abs().max()
    with pytest.raises(NotImplementedError):
        amd.HessianHook().add_batch(x.to(gpu))            # the kernel path takes fp16 / bf16 only
    # a 16-bit input through the fp32 branch gives the kernel path's H up to the order of the additions
    xh = x.half()
    a, b = amd.HessianHook(), amd.HessianHook()
    a.add_batch(xh.to(gpu), use_kernel=False)
    if shape[-1] % 4 == 0:
        b.add_batch(xh.to(gpu), use_kernel=True)
        assert (a.hessian - b.hessian).abs().max() <= 4e-6 * b.hessian.abs().max()


def test_reconstruct_nn_linear_places_result_like_the_reference(amd, gpu):
    """device defaults to the CPU (quant.py:262); the arithmetic still ran on the GPU and the weight
    is the same bits wherever it is put."""
    gen = torch.Generator().manual_seed(3)
    r, c, gsz = 64, 256, 128
    meta = {"qweight": torch.randint(0, 16, (r, c), generator=gen, dtype=torch.uint8),
            "scale": (torch.rand(r, c // gsz, generator=gen) * 0.01 + 0.002).half(),
            "qzero": torch.randint(0, 16, (r, c // gsz), generator=gen, dtype=torch.uint8),
            "perm_inv": torch.randperm(c, generator=gen).to(torch.int16),
            "group_sizes": torch.full([c // gsz], gsz, dtype=torch.int16),
            "group_bit_widths": torch.full([c // gsz], 4, dtype=torch.uint8)}
    on_cpu = amd.reconstruct_nn_linear(meta, dtype=torch.float16)
    on_gpu = amd.reconstruct_nn_linear(meta, dtype=torch.float16, device=gpu)
    assert on_cpu.weight.device.type == "cpu" and on_gpu.weight.device.type == "cuda"
    assert torch.equal(on_cpu.weight.data, on_gpu.weight.data.cpu())


@pytest.mark.parametrize("c", [1024, 1100, 1027])
def test_invert_applies_an_unstructured_order_exactly(amd, gpu, c):
    """The permuted, scaled Hessian that invert() leaves behind is H[perm][:, perm] * 2 / n + damp on the
    diagonal, bit for bit, for an activation order without structure (hidden sizes with and without 16-byte
    rows: the row-staged gather has a path for each)."""
    gen = torch.Generator(device=gpu).manual_seed(c)
    x = torch.randn(2 * c, c, generator=gen, device=gpu) * (0.5 + torch.rand(c, generator=gen, device=gpu))
    h = (x.t() @ x).contiguous()
    h = (h + h.t()) / 2
    hook = amd.HessianHook()
    hook.hessian, hook.n_samples = h.clone(), 4
    hook.invert(damp_ratio=1e-2, act_order=True)
    perm = hook.perm
    assert torch.equal(torch.sort(perm).values, torch.arange(c, device=gpu))
    assert int((perm[1:] - perm[:-1]).abs().max()) > c // 4          # no structure
    scaled = h * 0.5
    want = scaled[perm][:, perm].clone()
    assert torch.equal(want.diagonal(), torch.sort(scaled.diagonal(), descending=True).values)
    got = hook.hessian.clone()
    damp = got.diagonal() - want.diagonal()
    assert float(damp.min()) > 0 and float(damp.max() - damp.min()) <= 1e-6 * float(want.diagonal().max())
    got.diagonal().copy_(want.diagonal())
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_metric_sums_with_and_without_the_inverse_order(amd, gpu, dtype):
    """gq_quant_metrics reads W through perm or, given perm_inv, quant_t through perm_inv: the same sums
    (double accumulation; only the order of the additions differs)."""
    r, c = 200, 1100
    gen = torch.Generator(device=gpu).manual_seed(9)
    w = (torch.randn(r, c, generator=gen, device=gpu) * 0.05).to(dtype)
    perm = torch.randperm(c, generator=gen, device=gpu).to(torch.int32)
    perm_inv = torch.empty_like(perm)
    perm_inv[perm.long()] = torch.arange(c, device=gpu, dtype=torch.int32)
    quant_t = (w.float()[:, perm.long()] + 0.01 * torch.randn(r, c, generator=gen, device=gpu)).t().contiguous()
    a = amd.ops.quant_metrics(w, perm, None, quant_t, None)
    b = amd.ops.quant_metrics(w, perm, None, quant_t, None, perm_inv)
    exact = float(((quant_t.t().double() - w.double()[:, perm.long()]) ** 2).sum())
    assert abs(float(a[2]) - exact) <= 1e-12 * exact and abs(float(b[2]) - exact) <= 1e-12 * exact
    assert torch.equal(a[3], b[3]) or abs(float(a[3] - b[3])) <= 1e-12 * float(a[3])
    with pytest.raises(ValueError):
        amd.ops.quant_metrics(w, None, None, quant_t, None, perm_inv)


@pytest.mark.parametrize("r,c", [(8, 1024), (4, 2048), (64, 384)])
def test_quantize_layer_stays_inside_its_workspace(amd, gpu, r, c):
    """gq_quantize_layer with a column permutation, its workspace cut out of the middle of a buffer filled with a
    pattern: the bytes on both sides are untouched (few rows make every region of the workspace small next to
    the per-column scratch), and the levels equal those of a call with a workspace of its own."""
    from gptq_v2_amd import _lib
    from gptq_v2_amd import ops
    lib = _lib.load()
    gen = torch.Generator(device=gpu).manual_seed(r * c)
    w = (torch.randn(r, c, generator=gen, device=gpu) * 0.05).half()
    x = torch.randn(2 * c, c, generator=gen, device=gpu)
    hook = amd.HessianHook()
    hook.hessian, hook.n_samples = (x.t() @ x).contiguous(), 2
    u = hook.invert()
    perm = hook.perm.to(torch.int32).contiguous()
    gs = [128] * (c // 128)
    want = ops.quantize_layer(w, perm, u, gs, [4] * len(gs))
    need = lib.gq_quantize_layer_workspace(r, c)
    guard = 1 << 16
    buf = torch.full((guard + need + guard,), 0xA5, dtype=torch.uint8, device=gpu)
    qweight = torch.empty(r, c, dtype=torch.uint8, device=gpu)
    scale = torch.empty(r, len(gs), dtype=torch.float16, device=gpu)
    qzero = torch.empty(r, len(gs), dtype=torch.uint8, device=gpu)
    quant_t = torch.empty(c, r, device=gpu)
    error_t = torch.empty(c, r, device=gpu)
    arr = ops._int_array
    with torch.cuda.device(gpu):
        rc = lib.gq_quantize_layer(w.data_ptr(), ops._dtype_code(w), perm.data_ptr(), u.data_ptr(), r, c, arr(gs),
                                   arr([4] * len(gs)), len(gs), 0, ops.SEM_KERNEL, qweight.data_ptr(), scale.data_ptr(),
                                   qzero.data_ptr(), quant_t.data_ptr(), error_t.data_ptr(), buf.data_ptr() + guard, need,
                                   ops._stream(w))
    _lib.check(rc, "gq_quantize_layer")
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 0xA5).all()) and bool((buf[guard + need:] == 0xA5).all())
    assert torch.equal(qweight, want["qweight"]) and torch.equal(scale, want["scale"]) and torch.equal(qzero, want["qzero"])
