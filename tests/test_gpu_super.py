"""The two ways the native-loop column loop is launched -- one launch per 512 columns with the window in
LDS (quant_super_kernel, the default) and one launch per 128 columns with a GEMM between every two
(quant_block_kernel) -- give every output bit for bit: each weight receives the same fused multiply-adds
in the same order either way.  The C oracle is compared with both elsewhere (test_gpu_parity.py,
test_gpu_configs.py run the default; here the switch is flipped)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _factor(amd, c, n_tok, gpu, seed):
    gen = torch.Generator().manual_seed(seed)
    scales = (1 + 3 * torch.arange(c) / c)[torch.randperm(c, generator=gen)]
    hook = amd.HessianHook()
    hook.add_batch((torch.randn(1, n_tok, c, generator=gen) * scales).half().to(gpu))
    hook.invert()
    return hook, gen


CASES = [
    # rows, columns, group sizes, bits           what it exercises
    (64, 256, [128, 128], [4, 4]),               # one window, shorter than 512
    (256, 1024, [128] * 8, [4] * 8),             # two full windows
    (100, 1104, [128] * 8 + [80], [4] * 9),      # rows not a multiple of 32, a 80-column window tail, a 80-column group
    (32, 768, [16, 48, 64, 128, 256, 256], [3, 4, 2, 4, 8, 3]),   # groups of every multiple of 16, mixed bit widths
    (4096, 640, [128] * 5, [3] * 5),             # 128 workgroups
    (96, 512, [512], [4]),                       # one group as wide as the window
]


@pytest.mark.parametrize("r,c,gs,gb", CASES)
def test_super_block_launch_equals_block_launches(amd, gpu, r, c, gs, gb):
    from gptq_v2_amd import _lib, ops
    hook, gen = _factor(amd, c, max(2 * c, 512), gpu, seed=r * 31 + c)
    w = (torch.randn(r, c, generator=gen) * 0.02).half().to(gpu)
    w[0, :] = 0.0                       # a row whose grid is the epsilon alone
    w[min(3, r - 1), 5] = 300.0         # one outlier stretches a row's grid
    lib = _lib.load()
    before = lib.gq_set_quant_super(1)
    try:
        a = ops.quantize_layer(w, hook.perm.int(), hook.hessian_inv, gs, gb)
        lib.gq_set_quant_super(0)
        b = ops.quantize_layer(w, hook.perm.int(), hook.hessian_inv, gs, gb)
    finally:
        lib.gq_set_quant_super(before)
    for k in ("qweight", "scale", "qzero", "quant_t", "error_t"):
        assert torch.equal(a[k], b[k]), k
    assert int(a["qweight"].max()) <= max((1 << x) - 1 for x in gb)


def test_super_block_launch_without_the_optional_outputs(amd, gpu):
    """quant_t is optional; bf16 weights; no permutation."""
    from gptq_v2_amd import _lib, ops
    hook, gen = _factor(amd, 640, 1536, gpu, seed=7)
    w = (torch.randn(160, 640, generator=gen) * 0.02).to(torch.bfloat16).to(gpu)
    u = hook.hessian_inv
    lib = _lib.load()
    before = lib.gq_set_quant_super(1)
    try:
        a = ops.quantize_layer(w, None, u, [128] * 5, [4] * 5, want_quant=False)
        lib.gq_set_quant_super(0)
        b = ops.quantize_layer(w, None, u, [128] * 5, [4] * 5, want_quant=True)
    finally:
        lib.gq_set_quant_super(before)
    assert a["quant_t"] is None
    for k in ("qweight", "scale", "qzero", "error_t"):
        assert torch.equal(a[k], b[k]), k


def test_groups_that_are_not_multiples_of_16_take_the_block_kernel(amd, oracle, gpu):
    """A group of 40 columns: the layer falls back to one launch per block, whatever the switch says."""
    from gptq_v2_amd import ops
    hook, gen = _factor(amd, 168, 512, gpu, seed=9)
    w = (torch.randn(64, 168, generator=gen) * 0.02).half()
    gs, gb = [40, 128], [4, 4]
    got = ops.quantize_layer(w.to(gpu), hook.perm.int(), hook.hessian_inv, gs, gb)
    oh = oracle.OracleHessian()
    oh.hessian, oh.hessian_inv = hook.hessian.cpu(), hook.hessian_inv.cpu()
    oh.perm, oh.perm_inv = hook.perm.cpu(), hook.perm_inv.cpu()
    want = oracle.gptq_quant(w, oh, torch.tensor(gs, dtype=torch.int32), torch.tensor(gb, dtype=torch.int32),
                             use_kernel_semantics=True)["quant_meta"]
    assert torch.equal(got["qweight"].cpu(), want["qweight"])
