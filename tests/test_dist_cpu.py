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
    h = xs.t() @ xs
    n = gd.allreduce_hessian(h, e - b)
    assert n == 10
    full = x.reshape(-1, c)
    assert torch.allclose(h, full.t() @ full, rtol=1e-5, atol=1e-4)
    # every rank holds the same bits afterwards
    hs = [torch.empty_like(h) for _ in range(ws)]
    dist.all_gather(hs, h)
    assert all(torch.equal(hs[0], t) for t in hs)

    rb, re = gd.row_shard(n_rows, rank, ws, multiple=4)
    q = torch.arange(n_rows * 5, dtype=torch.uint8).reshape(n_rows, 5)
    s = torch.arange(n_rows * 2, dtype=torch.float16).reshape(n_rows, 2)
    got = gd.allgather_rows({"qweight": q[rb:re], "scale": s[rb:re]}, n_rows, multiple=4)
    assert torch.equal(got["qweight"], q) and torch.equal(got["scale"], s)
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
