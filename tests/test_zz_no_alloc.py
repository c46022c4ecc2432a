"""The allocation-free inversion entry point (gq_hessian_invert_static) against gq_hessian_invert:
same inputs through both, called directly through the C ABI, every output identical bit for bit.
Two calls in a row share the library's built-in workspace, the second on a smaller Hessian (whose
work matrices then overlay the remains of the first call's)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _hessian(c, seed, gpu):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(3 * c, c, generator=gen) * (1 + 3 * torch.arange(c) / c)
    return (x.t() @ x).float().to(gpu)


def _invert(lib, name, h, n_samples, act_order, stream):
    c = h.shape[0]
    h = h.clone()
    u = torch.full((c, c), float("nan"), device=h.device)
    perm = torch.full((c,), -1, dtype=torch.int32, device=h.device)
    perm_inv = torch.full((c,), -1, dtype=torch.int32, device=h.device)
    tries = ctypes.c_int(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    common = (p(h), p(u), p(perm), p(perm_inv), c, float(n_samples), 0.01, int(act_order), 100)
    if name == "static":
        rc = lib.gq_hessian_invert_static(*common, ctypes.byref(tries), stream)
    else:
        nbytes = lib.gq_hessian_invert_workspace(c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
        rc = lib.gq_hessian_invert(*common, p(ws), nbytes, ctypes.byref(tries), stream)
    assert rc == 0, lib.gq_last_error()
    return h, u, perm, perm_inv, tries.value


def test_static_workspace_inversion_equals_the_allocating_one(gpu):
    from gptq_v2_amd import _lib
    lib = _lib.load()
    assert lib.gq_hessian_invert_static_max_c() == 1024
    stream = ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    with torch.cuda.device(gpu):
        for c, act in ((1024, True), (768, True), (200, False), (1024, True)):
            h0 = _hessian(c, c + int(act), gpu)
            got = _invert(lib, "static", h0, 4, act, stream)
            want = _invert(lib, "plain", h0, 4, act, stream)
            assert got[4] == want[4] == 1
            assert torch.equal(got[0], want[0]), c                      # H as the call leaves it
            assert torch.equal(got[1], want[1]), c                      # U
            assert torch.isfinite(got[1]).all() and (got[1].diagonal() > 0).all()
            if act:
                assert torch.equal(got[2], want[2]) and torch.equal(got[3], want[3])
        # a Hessian that needs several dampings: the retry loop on the built-in workspace
        q, _ = torch.linalg.qr(torch.randn(256, 256, generator=torch.Generator().manual_seed(5), dtype=torch.float64))
        ev = torch.linspace(1.0, 2.0, 256, dtype=torch.float64)
        ev[0] = -0.0375
        hneg = ((q * ev) @ q.t())
        hneg = (0.25 * (hneg + hneg.t())).float().to(gpu)
        got = _invert(lib, "static", hneg, 1, False, stream)
        want = _invert(lib, "plain", hneg, 1, False, stream)
        assert got[4] == want[4] >= 3 and torch.equal(got[1], want[1])
        # larger than the built-in workspace: a clear error, nothing launched
        big = torch.eye(1152, device=gpu)
        tries = ctypes.c_int(0)
        rc = lib.gq_hessian_invert_static(ctypes.c_void_p(big.data_ptr()), ctypes.c_void_p(big.clone().data_ptr()), None, None,
                                          1152, 1.0, 0.01, 0, 100, ctypes.byref(tries), stream)
        assert rc == -1 and b"at most 1024" in lib.gq_last_error()
        torch.cuda.synchronize()


def test_static_workspace_inversion_allocates_no_device_memory(gpu):
    """What the entry point exists for: between a warm-up call and the calls that follow, the device's free memory
    does not move (hipMemGetInfo through torch.cuda.mem_get_info; torch's own caching allocator is kept out of it
    by preparing every tensor beforehand)."""
    from gptq_v2_amd import _lib
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(gpu):
        cases = []
        for c in (1024, 512, 1024):
            h = _hessian(c, 77 + c, gpu)
            cases.append((c, h, torch.empty(c, c, device=gpu), torch.empty(c, dtype=torch.int32, device=gpu),
                          torch.empty(c, dtype=torch.int32, device=gpu)))
        tries = ctypes.c_int(0)

        def call(c, h, u, perm, perm_inv):
            rc = lib.gq_hessian_invert_static(p(h), p(u), p(perm), p(perm_inv), c, 4.0, 0.01, 1, 100, ctypes.byref(tries), stream)
            assert rc == 0, lib.gq_last_error()

        call(*cases[0])                      # warm-up: code objects loaded, the symbol's address resolved
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info(gpu)
        for case in cases[1:]:
            call(*case)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info(gpu)
        assert free_after == free_before
