"""world_size-2 gloo test of the multi-GPU layout helpers (CPU, no kernels involved)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, ws, port, n_rows):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from gptq_v2_amd import dist as gd   # noqa: the package needs the built library, not a GPU
    torch.manual_seed(0)
    c = 32
    x = torch.randn(10, 6, c)                      # 10 calibration samples
    b, e = gd.sample_shard(10, rank, ws)
    xs = x[b:e].reshape(-1, c)
    full = x.reshape(-1, c)
    for algorithm, cc in (("direct", c), ("ring", c), ("direct", 30)):   # 30: a last row block that is not full
        h = xs[:, :cc].t() @ xs[:, :cc]
        h = torch.tril(h) + torch.tril(h, -1).t()      # exactly symmetric, as the XtX kernel leaves it
        n = gd.allreduce_hessian(h, e - b, algorithm=algorithm)
        assert n == 10
        assert torch.allclose(h, full[:, :cc].t() @ full[:, :cc], rtol=1e-5, atol=1e-4)
        assert torch.equal(h, h.t())
        # every rank holds the same bits afterwards
        hs = [torch.empty_like(h) for _ in range(ws)]
        dist.all_gather(hs, h)
        assert all(torch.equal(hs[0], t) for t in hs)

    rb, re = gd.row_shard(n_rows, rank, ws, multiple=4)
    q = torch.arange(n_rows * 5, dtype=torch.uint8).reshape(n_rows, 5)
    s = torch.arange(n_rows * 2, dtype=torch.float16).reshape(n_rows, 2)
    got = gd.allgather_rows({"qweight": q[rb:re], "scale": s[rb:re]}, n_rows, multiple=4)
    assert torch.equal(got["qweight"], q) and torch.equal(got["scale"], s)

    # a Linear with fewer 64-row units than ranks: rank 1's shard is empty, it contributes the
    # per-layer fields it can state itself and receives every row
    rows, cols, g = 64, 8, 2
    rb, re = gd.row_shard(rows, rank, ws)
    assert (re - rb) == (64 if rank == 0 else 0)
    fields = {"perm_inv": torch.arange(cols, dtype=torch.int16), "group_sizes": torch.tensor([4, 4], dtype=torch.int16),
              "group_bit_widths": torch.tensor([4, 4], dtype=torch.uint8), "qscale": None, "sscale": None,
              "scale_bit_width": None}
    local = None
    if re > rb:
        meta = dict(fields, qweight=torch.full((rows, cols), 7, dtype=torch.uint8),
                    scale=torch.full((rows, g), 0.5, dtype=torch.float16), qzero=torch.full((rows, g), 8, dtype=torch.uint8))
        local = {"quant_meta": meta, "metrics": {"gptq_error": 1.0, "gptq_norm": 2.0, "l2.0_error": 3.0, "l2.0_norm": 4.0}}
    res = gd.gather_quant_result(local, fields, rows, cols, torch.float16, "cpu")
    assert res["quant_meta"]["qweight"].shape == (rows, cols) and int(res["quant_meta"]["qweight"].sum()) == 7 * rows * cols
    assert torch.equal(res["quant_meta"]["perm_inv"], fields["perm_inv"])
    assert res["metrics"] == {"gptq_error": 1.0, "gptq_norm": 2.0, "l2.0_error": 3.0, "l2.0_norm": 4.0}

    # a set of two Linears of different heights in ONE all-gather: levels two to a byte, 16-bit scales, zero
    # points; metric sums in one all-reduce.  n_total spares the Hessian reduce its count collective.
    torch.manual_seed(3)
    heights, cols, g = [128, 192], 12, 3
    full = [{"qweight": torch.randint(0, 16, (h_, cols), dtype=torch.uint8), "scale": torch.rand(h_, g).half(),
             "qzero": torch.randint(0, 16, (h_, g), dtype=torch.uint8)} for h_ in heights]
    locals_, sums = [], torch.zeros(2, 4, dtype=torch.float64)
    for i, h_ in enumerate(heights):
        rb, re = gd.row_shard(h_, rank, ws)
        locals_.append({k: v[rb:re] for k, v in full[i].items()} if re > rb else None)
        sums[i] = torch.tensor([1.0, 2.0, 3.0, 4.0]) * (re - rb)
    parts, total = gd.gather_quant_set(locals_, sums, heights, cols, g, torch.float16, "cpu", bits=4)
    for i, h_ in enumerate(heights):
        for k in ("qweight", "scale", "qzero"):
            assert torch.equal(parts[i][k], full[i][k]), (i, k)
        assert total[i].tolist() == [1.0 * h_, 2.0 * h_, 3.0 * h_, 4.0 * h_]
    h = torch.eye(8) * (rank + 1)
    assert gd.allreduce_hessian(h, 5, n_total=10) == 10 and torch.equal(h, torch.eye(8) * 3)
    q8 = torch.randint(0, 256, (4, 6), dtype=torch.uint8)
    assert torch.equal(gd.unpack_levels(gd.pack_levels(q8, 8), 8, 6), q8) and gd.pack_levels(q8, 8).shape == (4, 6)
    q4 = torch.randint(0, 16, (4, 6), dtype=torch.uint8)
    assert gd.pack_levels(q4, 4).shape == (4, 3) and torch.equal(gd.unpack_levels(gd.pack_levels(q4, 4), 4, 6), q4)
    dist.barrier()
    dist.destroy_process_group()


def test_shards_cover_everything():
    from gptq_v2_amd import dist as gd
    for n, ws in ((128, 8), (10, 3), (5, 8), (0, 2)):
        spans = [gd.sample_shard(n, r, ws) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for n, ws, m in ((4096, 8, 64), (11008, 8, 64), (1024, 8, 64), (68, 2, 4), (64, 4, 64)):
        spans = [gd.row_shard(n, r, ws, m) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(b % m == 0 for b, _ in spans)


def test_allreduce_and_allgather_world_size_2():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 26), nprocs=2, join=True)


def test_pack_lower_roundtrip():
    from gptq_v2_amd import dist as gd
    for c in (4, 30, 32, 100, 257):
        a = torch.randn(c, c)
        h = torch.tril(a) + torch.tril(a, -1).t()
        flat = gd.pack_lower(h, multiple=8)
        assert flat.numel() % 8 == 0 and c * (c + 1) // 2 <= flat.numel() < c * (c + 1) // 2 + 8
        out = torch.full_like(h, float("nan"))
        gd.unpack_lower(flat, out)
        assert torch.equal(out, h)
