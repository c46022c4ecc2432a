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


def test_tail_workspace(amd):
    ws = amd.ops.xtx_tail_workspace
    assert ws(11008) == 4 * 178 * 256 * 256 * 4   # 946 lower tiles = 3 rounds of 256 + 178, those in 4 token runs
    assert ws(4096) == 0                          # 136 tiles: the uniform token split's case
    assert ws(8192) == 7 * 16 * 256 * 256 * 4     # 528 tiles = 2 rounds + 16: 7 runs, under half a round of short workgroups
    assert ws(9728) == 0                          # 741 tiles = 2 rounds + 229: the last round is nearly full already
    assert ws(11000) == 0                         # no 256-tiles
    assert ws(5888) > 0                           # 276 tiles = 1 round + 20


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c,ns", [(5888, (8192, 4096 + 19, 64)), (11008, (16384, 2048 + 5))])
def test_tail_accumulate_equals_single(amd, gpu, dtype, c, ns):
    """Calls of several lengths (one too short to cut, one that does not fill a k-step) into H + tail buffers, folded:
    the tiles of the whole rounds bit for bit those of the plain accumulation, the tail's equal up to the order of the
    fp32 sums, H exactly symmetric, buffers cleared, the same calls the same bits."""
    gen = torch.Generator(device=gpu).manual_seed(c)
    scale = 1 + 3 * torch.arange(c, device=gpu) / c
    xs = [(torch.randn(n, c, generator=gen, device=gpu) * scale).to(dtype) for n in ns]
    want = torch.zeros(c, c, device=gpu)
    for x in xs:
        amd.ops.xtx_accumulate(want, x)

    def run():
        h = torch.zeros(c, c, device=gpu)
        ws = torch.zeros(amd.ops.xtx_tail_workspace(c) // 4, device=gpu)
        for x in xs:
            amd.ops.xtx_accumulate_tail(h, ws, x)
        assert float(ws.abs().max()) > 0
        stale = int((h != want).sum())
        amd.ops.xtx_fold_tail(h, ws)
        assert float(ws.abs().max()) == 0.0
        return h, stale

    h, stale = run()
    assert stale > 0                                  # some tiles really went through the buffers
    assert torch.equal(h, h.t())
    differ = h != want
    nt = c // 256
    tiles_off = int(differ.view(nt, 256, nt, 256).any(dim=3).any(dim=1).sum())
    n_tail = amd.ops.xtx_tail_workspace(c) // (256 * 256 * 4)   # runs x tiles
    print(f"C={c}: {tiles_off} of {nt * nt} tiles differ in some bit from the plain accumulation")
    assert 0 < tiles_off <= 2 * n_tail
    assert float((h - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert torch.equal(run()[0], h)


def test_hook_uses_the_tail_split_at_11008(amd, gpu):
    c = 11008
    gen = torch.Generator(device=gpu).manual_seed(9)
    hook = amd.HessianHook()
    plain = torch.zeros(c, c, device=gpu)
    for _ in range(2):
        x = torch.randn(4, 2048, c, generator=gen, device=gpu).half()
        hook.add_batch(x)
        amd.ops.xtx_accumulate(plain, x.flatten(end_dim=-2))
    assert hook._tail is not None and hook._tail_dirty
    h = hook.hessian
    assert not hook._tail_dirty and torch.equal(h, h.t())
    assert float((h - plain).abs().max()) <= 1e-5 * float(plain.abs().max())
    # one sequence per call: not worth the buffers
    single = amd.HessianHook()
    single.add_batch(x[0])
    assert single._tail is None
    hook.hessian = torch.ones(1, device=gpu)
    assert hook._tail is None
