"""The dense replay GEMM (gq_gemm_nt: y = x W^T on a reconstructed weight, the reference's
`lin(h)` at llama.py:176-177): against the fp64 product of the same 16-bit operands, within the
tolerance of fp32 accumulation and one rounding of the result to the 16-bit type."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tol(dtype):
    # one rounding to the output type (2^-11 for fp16, 2^-8 for bf16) plus fp32 accumulation
    return 1e-3 if dtype == torch.float16 else 6e-3


@pytest.mark.parametrize("m,n,k,dtype", [
    (256, 256, 64, torch.float16), (300, 200, 128, torch.float16), (1, 5, 64, torch.float16),
    (513, 770, 192, torch.bfloat16), (1000, 1027, 256, torch.float16), (2048, 4096, 4096, torch.float16),
    (777, 11008, 4096, torch.bfloat16), (4096, 4096, 11008, torch.float16),
])
def test_gemm_nt_against_fp64(amd, gpu, m, n, k, dtype):
    """Ragged row and column counts (tiles that are not full, output widths that are not multiples
    of 4 and take the narrow stores), one to 172 k-steps, fp16 and bf16."""
    gen = torch.Generator(device=gpu).manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=gen, device=gpu) * (1 + 3 * torch.arange(k, device=gpu) / k)).to(dtype)
    w = (torch.randn(n, k, generator=gen, device=gpu) * 0.02).to(dtype)
    y = amd.ops.gemm_nt(x, w)
    assert y.shape == (m, n) and y.dtype == dtype
    if m * n * k <= 2 ** 32:
        ref = x.double() @ w.double().t()
        assert (y.double() - ref).abs().max() <= _tol(dtype) * ref.abs().max()
    else:
        rows = torch.cat([torch.tensor([0, 255, 256, m - 1], device=gpu), torch.randint(0, m, (124,), generator=gen, device=gpu)])
        cols = torch.cat([torch.tensor([0, 255, 256, n - 1], device=gpu), torch.randint(0, n, (252,), generator=gen, device=gpu)])
        ref = x[rows].double() @ w[cols].double().t()
        assert (y[rows][:, cols].double() - ref).abs().max() <= _tol(dtype) * ref.abs().max()
        # every row and column was written: column sums against those of the fp64 product
        ref_colsum = x.double().sum(dim=0) @ w.double().t()
        assert (y.double().sum(dim=0) - ref_colsum).abs().max() <= _tol(dtype) * ref.abs().max() * m ** 0.5


def test_gemm_nt_identity_and_rejections(amd, gpu):
    """x = I picks rows of W exactly (each output is one product plus zeros); empty input; shapes the
    kernel does not take are refused with an error, not computed wrongly."""
    k = 256
    w = (torch.randn(300, k, device=gpu) * 0.1).half()
    eye = torch.eye(k, device=gpu).half()
    assert torch.equal(amd.ops.gemm_nt(eye, w), w.t())
    assert amd.ops.gemm_nt(torch.empty(0, k, device=gpu, dtype=torch.float16), w).shape == (0, 300)
    with pytest.raises(NotImplementedError):
        amd.ops.gemm_nt(torch.zeros(4, 100, device=gpu, dtype=torch.float16), torch.zeros(8, 100, device=gpu, dtype=torch.float16))
    with pytest.raises(ValueError):
        amd.ops.gemm_nt(torch.zeros(4, 128, device=gpu, dtype=torch.float16), torch.zeros(8, 64, device=gpu, dtype=torch.float16))
    with pytest.raises(ValueError):
        amd.ops.gemm_nt(torch.zeros(4, 128, device=gpu, dtype=torch.float16), torch.zeros(8, 128, device=gpu, dtype=torch.bfloat16))


def test_hip_linear_is_the_replay_of_the_packed_layer(amd, gpu):
    """reconstruct_nn_linear(meta)(x) -- the dense replay -- and QuantLinear(meta)(x) -- the packed
    layer -- multiply the same dequantised numbers with fp32 accumulation: they agree to the rounding
    of the result, with act-order, on a q_proj-sized layer; and HipLinear is an nn.Linear with the
    weight the reference would hold."""
    gen = torch.Generator(device=gpu).manual_seed(5)
    r, c, gsz, m = 4096, 4096, 128, 1024
    meta = {"qweight": torch.randint(0, 16, (r, c), generator=gen, device=gpu, dtype=torch.uint8),
            "scale": (torch.rand(r, c // gsz, generator=gen, device=gpu) * 0.01 + 0.002).half(),
            "qzero": torch.randint(0, 16, (r, c // gsz), generator=gen, device=gpu, dtype=torch.uint8),
            "perm_inv": torch.argsort(torch.randperm(c, generator=gen, device=gpu)).to(torch.int16),
            "group_sizes": torch.full([c // gsz], gsz, dtype=torch.int16),
            "group_bit_widths": torch.full([c // gsz], 4, dtype=torch.uint8)}
    lin = amd.reconstruct_nn_linear(meta, torch.float16, gpu)
    assert isinstance(lin, torch.nn.Linear) and lin.bias is None
    assert torch.equal(lin.weight.data, amd.quant.reconstruct_weight(meta, torch.float16, gpu))
    packed = amd.QuantLinear.from_quant_meta(meta, gpu)
    x = torch.randn(2, m // 2, c, generator=gen, device=gpu).half()
    y_dense, y_packed = lin(x), packed(x)
    assert y_dense.shape == y_packed.shape == (2, m // 2, r)
    scale = float(y_dense.float().abs().max())
    assert float((y_dense.float() - y_packed.float()).abs().max()) <= 2e-3 * scale
    ref = torch.nn.functional.linear(x.double(), lin.weight.data.double())
    assert float((y_dense.double() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_nt_rows_do_not_depend_on_the_token_count(amd, gpu, dtype):
    """The rows of a small product (few tiles, a partly empty persistent grid, ragged last tile) equal the same
    rows inside a large one bit for bit: an output element is one sum in one order whatever the tiling."""
    gen = torch.Generator(device=gpu).manual_seed(11)
    n, k = 4096 + 24, 1024
    x = torch.randn(8192, k, generator=gen, device=gpu).to(dtype)
    w = (torch.randn(n, k, generator=gen, device=gpu) * 0.05).to(dtype)
    big = amd.ops.gemm_nt(x, w)                 # 32 x 17 tiles of 256 x 256
    for m in (2048, 1000, 130):
        assert torch.equal(amd.ops.gemm_nt(x[:m].contiguous(), w), big[:m])
