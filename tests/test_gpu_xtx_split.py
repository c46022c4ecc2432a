"""H += X^T X with the tokens of a call cut into runs that accumulate into partial Hessians
(gq_xtx_accumulate_split / gq_xtx_fold) and HessianHook's use of it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_split_hint(amd):
    hint = amd.ops.xtx_split_hint
    assert hint(4096, 32768) == 3        # 136 lower tiles on 256 CUs: thirds give two full rounds
    assert hint(4096, 2048) == 1         # one sequence per call: the extra read-modify-writes would dominate
    assert hint(11008, 32768) == 1       # 946 tiles fill the chip
    assert hint(4100, 32768) == 1        # not a multiple of the 256-tile
    assert hint(256, 1 << 20) > 1
    assert hint(4096, 0) == 1


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c,n,parts", [(512, 4096 + 17, 3), (1024, 3000, 4), (256, 96, 2), (512, 64, 3), (384, 640, 2)])
def test_split_accumulate_equals_single(amd, gpu, dtype, c, n, parts):
    """Two calls into S parts, folded: equal to the unsplit accumulation up to the order of the fp32 sums,
    exactly symmetric, extra parts cleared; a token count that does not fill a k-step, fewer k-steps than
    parts and a hidden size without 256-tiles (everything into part 0) included."""
    gen = torch.Generator().manual_seed(c + n)
    x1 = torch.randn(n, c, generator=gen).to(dtype).to(gpu)
    x2 = torch.randn(n // 2 + 3, c, generator=gen).to(dtype).to(gpu)
    want = torch.zeros(c, c, device=gpu)
    amd.ops.xtx_accumulate(want, x1)
    amd.ops.xtx_accumulate(want, x2)
    p = torch.zeros(parts, c, c, device=gpu)
    amd.ops.xtx_accumulate_split(p, x1)
    amd.ops.xtx_accumulate_split(p, x2)
    if c % 256 == 0 and n // 32 >= parts:
        assert float(p[1].abs().max()) > 0    # the split really happened
    exact = x1.double().t() @ x1.double() + x2.double().t() @ x2.double()
    amd.ops.xtx_fold(p)
    h = p[0]
    assert torch.equal(h, h.t())
    assert float(p[1:].abs().max()) == 0.0
    tol = 1e-5 * float(exact.abs().max())
    assert float((h.double() - exact).abs().max()) <= tol
    assert float((want.double() - exact).abs().max()) <= tol
    # deterministic: the same calls give the same bits
    p2 = torch.zeros(parts, c, c, device=gpu)
    amd.ops.xtx_accumulate_split(p2, x1)
    amd.ops.xtx_accumulate_split(p2, x2)
    amd.ops.xtx_fold(p2)
    assert torch.equal(p2[0], h)


def test_hook_splits_at_4096_and_matches_unsplit(amd, gpu):
    """HessianHook at C = 4096 with 16-sequence batches accumulates into three parts; `hessian` reads the
    folded sum; invert() and the rest see an ordinary (C, C) tensor."""
    c = 4096
    gen = torch.Generator(device=gpu).manual_seed(3)
    hook = amd.HessianHook()
    plain = torch.zeros(c, c, device=gpu)
    for _ in range(2):
        x = (torch.randn(16, 2048, c, generator=gen, device=gpu) * (1 + 3 * torch.arange(c, device=gpu) / c)).half()
        hook.add_batch(x)
        amd.ops.xtx_accumulate(plain, x.flatten(end_dim=-2))
    assert hook._parts is not None and hook._parts.shape == (3, c, c) and hook._parts_dirty
    h = hook.hessian
    assert not hook._parts_dirty and h.shape == (c, c) and h.data_ptr() == hook._parts.data_ptr()
    assert torch.equal(h, h.t())
    assert float((h - plain).abs().max()) <= 1e-5 * float(plain.abs().max())   # fp32 sums of 65536 products in two orders
    assert hook.n_samples == 32
    u = hook.invert()
    assert u.shape == (c, c) and bool(torch.isfinite(u).all())
    # a batch of one sequence does not split
    hook2 = amd.HessianHook()
    hook2.add_batch(x[0])
    assert hook2._parts is None
    # assigning the Hessian drops the parts
    hook3 = amd.HessianHook()
    hook3.add_batch(x)
    hook3.hessian = torch.ones(1, device=gpu)
    assert hook3._parts is None and hook3.hessian.numel() == 1


def test_hook_with_two_accumulators_matches_one(amd, gpu):
    """overlap_batches: calls alternate between two accumulators on two streams where a call is a short
    launch (C = 4096), and nowhere else.  The sum is taken in a fixed order: the same calls give the same
    bits, equal to the one-accumulator Hessian up to the order of the fp32 sums; an input may be freed
    right after the call; the caller's stream sees the finished sum when it reads `hessian`."""
    c = 4096
    scale = 1 + 3 * torch.arange(c, device=gpu) / c

    def run(overlap, seed=5):
        gen = torch.Generator(device=gpu).manual_seed(seed)
        hook = amd.HessianHook(overlap_batches=overlap)
        for _ in range(5):    # an odd count: the accumulators get three and two calls
            x = (torch.randn(8, 2048, c, generator=gen, device=gpu) * scale).half()
            hook.add_batch(x)
            del x             # freed while the side stream may still read it
        return hook

    two, one = run(True), run(False)
    assert two._lanes == 2 and two._parts.shape == (6, c, c) and one._lanes == 1
    h2, h1 = two.hessian.clone(), one.hessian
    assert torch.equal(h2, h2.t())
    assert float(two._parts[1:].abs().max()) == 0.0
    assert float((h2 - h1).abs().max()) <= 1e-5 * float(h1.abs().max())
    assert two.n_samples == one.n_samples == 40
    assert torch.equal(run(True).hessian, h2)
    u2, u1 = two.invert(), one.invert()
    assert torch.equal(two.perm, one.perm)
    assert float((u2 - u1).abs().max()) <= 1e-3 * float(u1.abs().max())
    # long launches (C = 11008 fills the chip several times) and short inputs keep one accumulator
    wide = amd.HessianHook(overlap_batches=True)
    wide.add_batch(torch.zeros(8, 2048, 11008, dtype=torch.float16, device=gpu))
    assert wide._lanes == 1
    short = amd.HessianHook(overlap_batches=True)
    short.add_batch(torch.zeros(2048, c, dtype=torch.float16, device=gpu))
    assert short._lanes == 1
