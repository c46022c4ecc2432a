"""The widened packed path: layers with mixed group sizes / bit widths / EXL2 scales through
QuantLinear, dense 3-bit storage, and the stream-structured packed GEMM against the split one."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import load_golden, t

pytestmark = pytest.mark.gpu


def _meta(g):
    meta = {"qweight": t(g["qweight"]), "scale": t(g["scale"]), "qzero": t(g["qzero"]), "perm_inv": t(g["perm_inv"]),
            "group_sizes": t(g["meta_group_sizes"]), "group_bit_widths": t(g["meta_group_bit_widths"])}
    for k in ("qscale", "sscale"):
        if k in g:
            meta[k] = t(g[k])
    return meta


def test_quantlinear_mixed_groups_exl2_fixture(amd, oracle, gpu):
    """The committed EXL2 fixture (groups of 64/64/128 columns at 4/3/2... bits as the fixture holds them,
    scales qscale^2 * sscale): unpack() returns quant_meta bit for bit, identity input returns the dense
    reconstruct_nn_linear weight bit for bit, a random input matches the fp64 contraction."""
    g = load_golden("f8_kernelsem_exl2_64x256_mixed")
    meta = _meta(g)
    assert len(set(meta["group_sizes"].tolist())) > 1 or len(set(meta["group_bit_widths"].tolist())) > 1
    layer = amd.QuantLinear.from_quant_meta(meta, gpu)
    back = layer.unpack()
    for k in ("qweight", "scale", "qzero", "perm_inv", "group_sizes", "group_bit_widths", "qscale", "sscale"):
        if meta.get(k) is not None:
            assert torch.equal(back[k].cpu(), meta[k]), k
    lin = amd.reconstruct_nn_linear(meta, dtype=torch.float16, device=gpu)
    want_w = oracle.construct_weight(meta, torch.float16)
    assert torch.equal(lin.weight.data.cpu(), want_w)
    if "recon_weight" in g:
        assert torch.equal(want_w, t(g["recon_weight"]).to(torch.float16))
    c = meta["qweight"].shape[1]
    assert torch.equal(layer(torch.eye(c, dtype=torch.float16, device=gpu)).cpu(), want_w.t())
    x = torch.randn(5, 40, c, generator=torch.Generator().manual_seed(4)).half()
    y = layer(x.to(gpu)).cpu()
    want = oracle.w4a16_linear(x.reshape(-1, c), meta).reshape(5, 40, -1)
    assert torch.allclose(y.float(), want, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_quantlinear_mixed_group_sizes_and_widths_random(amd, gpu, dtype):
    """Groups of 32 / 96 / 128 / 64 / ... columns (finest common grid 32) at 2, 3 and 4 bits with act-order."""
    gen = torch.Generator().manual_seed(77)
    r, sizes = 300, [32, 96, 128, 64, 64, 128]
    widths = [4, 3, 2, 4, 3, 4]
    c = sum(sizes)
    q = torch.cat([torch.randint(0, 1 << b, (r, s), generator=gen, dtype=torch.uint8) for s, b in zip(sizes, widths)], 1).to(gpu)
    zero = torch.stack([torch.randint(0, 1 << b, (r,), generator=gen, dtype=torch.uint8) for b in widths], 1).to(gpu)
    scale = (torch.rand(r, len(sizes), generator=gen) * 0.01 + 0.002).to(dtype).to(gpu)
    perm_inv = torch.randperm(c, generator=gen).to(torch.int16)
    meta = {"qweight": q, "scale": scale, "qzero": zero, "qscale": None, "sscale": None, "perm_inv": perm_inv,
            "group_sizes": torch.tensor(sizes, dtype=torch.int16), "group_bit_widths": torch.tensor(widths, dtype=torch.uint8)}
    layer = amd.QuantLinear.from_quant_meta(meta, gpu)
    assert layer.group_size == 32 and layer.scales.shape == (c // 32, r)
    w = amd.reconstruct_nn_linear(meta, dtype=dtype, device=gpu).weight.data
    assert torch.equal(layer(torch.eye(c, dtype=dtype, device=gpu)), w.t())
    back = layer.unpack()
    assert torch.equal(back["qweight"], q) and torch.equal(back["scale"], scale) and torch.equal(back["qzero"], zero)
    assert back["group_sizes"].tolist() == sizes and back["group_bit_widths"].tolist() == widths
    assert torch.equal(back["perm_inv"].cpu(), perm_inv)
    bad = dict(meta, group_sizes=torch.tensor([48, 80, 128, 64, 64, 128], dtype=torch.int16))
    with pytest.raises(NotImplementedError):   # no common divisor that is a multiple of 32
        amd.QuantLinear.from_quant_meta(bad, gpu)


@pytest.mark.parametrize("n", [64, 192, 4096])
def test_pack_w3_kblock_matches_oracle_and_roundtrips(amd, oracle, gpu, n):
    gen = torch.Generator().manual_seed(n)
    v = torch.randint(0, 8, (9, n), generator=gen, dtype=torch.uint8)
    p = amd.ops.pack_rows(v.to(gpu), 3, amd.ops.PACK_W3_KBLOCK)
    assert p.shape == (n // 64, 9, 6)
    assert torch.equal(p.cpu(), oracle.pack_rows_w3_kblock(v))
    assert torch.equal(amd.ops.unpack_rows(p, 3, n, amd.ops.PACK_W3_KBLOCK).cpu(), v)
    with pytest.raises(ValueError):
        amd.ops.pack_rows(v.to(gpu), 4, amd.ops.PACK_W3_KBLOCK)


def _random_layer(amd, gpu, r, c, gsz, bits, dtype, seed, act_order=False):
    gen = torch.Generator().manual_seed(seed)
    q = torch.randint(0, 1 << bits, (r, c), generator=gen, dtype=torch.uint8).to(gpu)
    scale = (torch.rand(r, c // gsz, generator=gen) * 0.01 + 0.002).to(dtype).to(gpu)
    zero = torch.randint(0, 1 << bits, (r, c // gsz), generator=gen, dtype=torch.uint8).to(gpu)
    perm_inv = torch.randperm(c, generator=gen).to(torch.int16) if act_order else None
    return {"qweight": q, "scale": scale, "qzero": zero, "qscale": None, "sscale": None, "perm_inv": perm_inv,
            "group_sizes": torch.full([c // gsz], gsz, dtype=torch.int16), "group_bit_widths": torch.full([c // gsz], bits, dtype=torch.uint8)}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("r,c,gsz", [(320, 512, 128), (4352, 4352, 128), (300, 192, 32), (513, 64, 64), (700, 128, 64),
                                     (600, 384, 96), (260, 1024, 256)])
def test_dense3_identity_is_dequantised_weight(amd, gpu, dtype, r, c, gsz):
    """3 bits per weight in memory: identity input returns the dense weight bit for bit (every k-step
    count class of the stream kernel: 1, 2, 3, even, odd), unpack() restores quant_meta."""
    meta = _random_layer(amd, gpu, r, c, gsz, 3, dtype, r + c, act_order=True)
    w = amd.reconstruct_nn_linear(meta, dtype=dtype, device=gpu).weight.data
    layer = amd.QuantLinear.from_quant_meta(meta, gpu, dense3=True)
    assert layer.qweight.shape == (c // 64, r, 6)
    assert torch.equal(layer(torch.eye(c, dtype=dtype, device=gpu)), w.t())
    back = layer.unpack()
    assert torch.equal(back["qweight"], meta["qweight"]) and torch.equal(back["qzero"], meta["qzero"])
    with pytest.raises(ValueError):
        amd.QuantLinear.from_quant_meta(_random_layer(amd, gpu, 64, 64, 64, 4, dtype, 1), gpu, dense3=True)


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,r,c,gsz", [(2048, 11008, 4096, 128), (300, 520, 4096, 128), (4099, 4100, 320, 32), (1, 1, 64, 64),
                                       (16384, 4096, 4096, 128)])
def test_stream_kernel_products(amd, gpu, bits, dtype, m, r, c, gsz):
    """gq_wq_gemm's stream kernel (forced for 4 bits, the only one for dense 3 bits) against the fp64
    contraction with the dense weight, ragged row / column counts and a single element included."""
    lib = amd._lib.load()
    meta = _random_layer(amd, gpu, r, c, gsz, bits, dtype, m + r + c + bits)
    w = amd.quant.reconstruct_weight(meta, dtype, gpu)
    packed = amd.ops.pack_rows(meta["qweight"], bits, amd.ops.PACK_W4_KBLOCK if bits == 4 else amd.ops.PACK_W3_KBLOCK)
    scale_t, zero_t = meta["scale"].t().contiguous(), meta["qzero"].t().contiguous()
    x = torch.randn(m, c, generator=torch.Generator().manual_seed(5)).to(dtype).to(gpu)
    before = lib.gq_set_wq_kernel(1)
    try:
        y = amd.ops.w4a16_gemm(x, packed, scale_t, zero_t, gsz, bits=bits)
    finally:
        lib.gq_set_wq_kernel(before)
    rows = torch.randperm(m, device=gpu)[:64]
    cols = torch.randperm(r, device=gpu)[:256]
    ref = x[rows].double() @ w[cols].double().t()
    got = y[rows][:, cols].double()
    tol = 2e-3 if dtype == torch.float16 else 2 ** -7
    assert (got - ref).abs().max() <= tol * ref.abs().max() + 1e-3
    if bits == 4:
        lib.gq_set_wq_kernel(2)
        try:
            y2 = amd.ops.w4a16_gemm(x, packed, scale_t, zero_t, gsz)
        finally:
            lib.gq_set_wq_kernel(before)
        assert (y2.double() - y.double()).abs().max() <= tol * ref.abs().max() + 1e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,r,c,gsz", [(2048, 4096, 4096, 128), (300, 520, 4096, 128), (4099, 4100, 320, 64), (256, 128, 64, 64),
                                       (515, 129, 192, 64)])
def test_split_kernel_every_tile_shape(amd, gpu, dtype, m, r, c, gsz):
    """The split 4-bit kernel with each tile shape forced (128 x 128, 256 x 256, 256 tokens x 128 rows): the same
    product bit for bit -- every output element is the same k-ascending sum whatever the tile -- and, with an
    identity-like input, the dense weight itself; ragged token and row counts, one to 64 k-steps."""
    lib = amd._lib.load()
    meta = _random_layer(amd, gpu, r, c, gsz, 4, dtype, m + r + c)
    w = amd.quant.reconstruct_weight(meta, dtype, gpu)
    packed = amd.ops.pack_rows(meta["qweight"], 4, amd.ops.PACK_W4_KBLOCK)
    scale_t, zero_t = meta["scale"].t().contiguous(), meta["qzero"].t().contiguous()
    x = torch.randn(m, c, generator=torch.Generator().manual_seed(7)).to(dtype).to(gpu)
    eye = torch.zeros(max(m, c), c, dtype=dtype, device=gpu)
    eye[:c] = torch.eye(c, dtype=dtype, device=gpu)
    kernel_before, tile_before = lib.gq_set_wq_kernel(2), lib.gq_set_w4_tile(128)
    try:
        ys, ws = {}, {}
        for tile in (128, 256, 384):
            lib.gq_set_w4_tile(tile)
            ys[tile] = amd.ops.w4a16_gemm(x, packed, scale_t, zero_t, gsz)
            ws[tile] = amd.ops.w4a16_gemm(eye, packed, scale_t, zero_t, gsz)
    finally:
        lib.gq_set_w4_tile(tile_before)
        lib.gq_set_wq_kernel(kernel_before)
    for tile in (128, 256, 384):
        assert torch.equal(ws[tile][:c], w.t()), tile
        assert float(ws[tile][c:].abs().max()) == 0.0 if m > c else True
    assert torch.equal(ys[256], ys[384])
    assert torch.equal(ys[128], ys[384])
    assert lib.gq_set_w4_tile(7) == -1


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_stream_kernel_every_k_step_count(amd, gpu, bits, dtype):
    """1 to 12 k-steps (the stream kernel peels its first and its last two to four steps, by parity): the
    product with an identity-like input returns the dense weight bit for bit at every count."""
    lib = amd._lib.load()
    before = lib.gq_set_wq_kernel(1)
    try:
        for steps in range(1, 13):
            c, r, gsz = 64 * steps, 264, 32 if steps % 2 else 64
            meta = _random_layer(amd, gpu, r, c, gsz, bits, dtype, 100 * steps + bits)
            w = amd.quant.reconstruct_weight(meta, dtype, gpu)
            packed = amd.ops.pack_rows(meta["qweight"], bits, amd.ops.PACK_W4_KBLOCK if bits == 4 else amd.ops.PACK_W3_KBLOCK)
            y = amd.ops.w4a16_gemm(torch.eye(c, dtype=dtype, device=gpu), packed, meta["scale"].t().contiguous(),
                                   meta["qzero"].t().contiguous(), gsz, bits=bits)
            assert torch.equal(y, w.t()), steps
    finally:
        lib.gq_set_wq_kernel(before)


def _random_meta(gen, gpu, r, c, group, bits, dtype):
    return {"qweight": torch.randint(0, 1 << bits, (r, c), generator=gen, dtype=torch.uint8).to(gpu),
            "scale": (torch.rand(r, c // group, generator=gen) * 0.01 + 0.002).to(dtype).to(gpu),
            "qzero": torch.randint(0, 1 << bits, (r, c // group), generator=gen, dtype=torch.uint8).to(gpu),
            "qscale": None, "sscale": None, "perm_inv": torch.randperm(c, generator=gen).to(torch.int16),
            "group_sizes": torch.full([c // group], group, dtype=torch.int16),
            "group_bit_widths": torch.full([c // group], bits, dtype=torch.uint8)}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bits,group", [(4, 128), (4, 32), (3, 64)])
def test_expand_once_equals_reconstruction_and_packed_forward(amd, gpu, dtype, bits, group):
    """gq_wq_expand writes the dense weight of a packed operand: equal to reconstruct_nn_linear's weight with
    the columns in act-order, bit for bit; the forward that expands once and runs the dense kernel returns the
    packed kernel's product bit for bit (ragged token and row counts included)."""
    gen = torch.Generator().manual_seed(1000 * bits + group)
    r, c = 8192 + 40, 384
    meta = _random_meta(gen, gpu, r, c, group, bits, dtype)
    layer = amd.QuantLinear.from_quant_meta(meta, gpu, dense3=bits == 3)
    dense = amd.ops.wq_expand(layer.qweight, layer.scales, layer.qzeros, layer.group_size, bits=3 if layer.dense3 else 4)
    w = amd.reconstruct_nn_linear(meta, dtype=dtype, device=gpu).weight.data
    assert torch.equal(dense, w[:, layer.perm.long()])
    x = torch.randn(300, c, generator=gen).to(dtype).to(gpu)
    layer.expand_min_tokens = 0
    packed = layer(x)
    layer.expand_min_tokens = 256
    assert layer._expands(x.shape[0])
    assert torch.equal(layer(x), packed)
    with pytest.raises(ValueError):
        amd.ops.wq_expand(layer.qweight, layer.scales, layer.qzeros, layer.group_size, bits=3 if layer.dense3 else 4,
                          out=torch.empty(r, c + 64, dtype=dtype, device=gpu))


def test_quantlinear_concat_returns_the_parts_side_by_side(amd, gpu):
    """q/k/v as one layer: one gather, one launch, outputs equal to the three layers' own bit for bit, with the
    packed kernel and with the expand-once path."""
    gen = torch.Generator().manual_seed(99)
    c, rows = 512, [256, 128, 320]
    perm_inv = torch.randperm(c, generator=gen).to(torch.int16)
    parts = []
    for r in rows:
        meta = _random_meta(gen, gpu, r, c, 128, 4, torch.float16)
        meta["perm_inv"] = perm_inv
        parts.append(amd.QuantLinear.from_quant_meta(meta, gpu))
    fused = amd.QuantLinear.concat(parts)
    assert fused.out_features == sum(rows) and fused.qweight.shape == (c // 64, sum(rows), 8)
    x = torch.randn(3, 100, c, generator=gen).half().to(gpu)
    for least in (0, 64):
        fused.expand_min_tokens = least
        ys = fused(x).split(rows, dim=-1)
        for part, y in zip(parts, ys):
            part.expand_min_tokens = 0
            assert torch.equal(part(x), y)
    other = amd.QuantLinear.from_quant_meta(_random_meta(gen, gpu, 64, c, 128, 4, torch.float16), gpu)   # its own act-order
    with pytest.raises(ValueError):
        amd.QuantLinear.concat([parts[0], other])
