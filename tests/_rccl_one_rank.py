"""Helper of test_gpu_dist.py (not collected): a process group of ONE rank on the RCCL backend, with
the collectives forced on, so that the device code path of gptq_v2_amd.dist runs on a one-GPU box."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import gptq_v2_amd as amd
    from gptq_v2_amd import dist as gd
    gd.FORCE_COLLECTIVES = True
    gen = torch.Generator(device=dev).manual_seed(0)
    for c in (384, 1000, 4096):
        x = torch.randn(2, 512, c, generator=gen, device=dev).half()
        hook = amd.HessianHook()
        hook.add_batch(x)
        before = hook.hessian.clone()
        for algorithm in ("direct", "ring"):
            n = gd.allreduce_hessian(hook.hessian, hook.n_samples, algorithm=algorithm)
            assert n == 2
            assert torch.equal(hook.hessian, before), (c, algorithm)   # one rank: the sum is the matrix itself, bit for bit
    rows = 200
    parts = {"qweight": torch.randint(0, 16, (rows, 64), generator=gen, device=dev, dtype=torch.uint8),
             "scale": torch.rand(rows, 2, generator=gen, device=dev).half()}
    got = gd.allgather_rows(parts, rows)
    assert all(torch.equal(got[k], parts[k]) for k in parts)
    fields = {"perm_inv": torch.arange(64, dtype=torch.int16, device=dev), "group_sizes": torch.tensor([32, 32], dtype=torch.int16),
              "group_bit_widths": torch.tensor([4, 4], dtype=torch.uint8), "qscale": None, "sscale": None, "scale_bit_width": None}
    meta = dict(fields, qzero=torch.zeros(rows, 2, dtype=torch.uint8, device=dev), **parts)
    res = gd.gather_quant_result({"quant_meta": meta, "metrics": {"gptq_error": 1.5, "gptq_norm": 2.5, "l2.0_error": 3.5,
                                                                  "l2.0_norm": 0.25}}, fields, rows, 64, torch.float16, dev)
    assert torch.equal(res["quant_meta"]["qweight"], parts["qweight"]) and res["metrics"]["gptq_error"] == 1.5
    # the triangle kernels on their own: pack -> unpack restores a symmetric matrix bit for bit, whatever the size
    for c in (5, 64, 100, 1000):
        a = torch.randn(c, c, generator=gen, device=dev)
        h = torch.tril(a) + torch.tril(a, -1).t()
        flat = gd.pack_lower(h, multiple=8)
        assert flat.numel() % 8 == 0 and flat.numel() - c * (c + 1) // 2 < 8
        r_, c_ = torch.tril_indices(c, c, device=dev)
        assert torch.equal(flat[: c * (c + 1) // 2], h[r_, c_])
        out = torch.full_like(h, float("nan"))
        gd.unpack_lower(flat, out)
        assert torch.equal(out, h)
    # the per-set gather on device tensors
    heights = [128, 200]
    full = [{"qweight": torch.randint(0, 16, (h_, 64), generator=gen, device=dev, dtype=torch.uint8),
             "scale": torch.rand(h_, 2, generator=gen, device=dev).half(),
             "qzero": torch.randint(0, 16, (h_, 2), generator=gen, device=dev, dtype=torch.uint8)} for h_ in heights]
    sums = torch.arange(8, dtype=torch.float64, device=dev).reshape(2, 4)
    got_parts, got_sums = gd.gather_quant_set(full, sums.clone(), heights, 64, 2, torch.float16, dev, bits=4)
    assert all(torch.equal(got_parts[i][k], full[i][k]) for i in range(2) for k in full[i]) and torch.equal(got_sums, sums)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("rccl one-rank ok")


if __name__ == "__main__":
    main()
