"""magma_spd_inverse_static: the entry point that allocates nothing (static device workspace,
N <= 1024) against magma_spd_inverse on the same inputs -- same kernels, so bit for bit -- on two
calls in a row, the second smaller (the workspace keeps the first call's contents), then a third
larger again; and the size limit. Collected last on purpose (file name)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def _problem(torch, lib, N, seed):
    Np = lib.magma_pad(N)
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, 10, N))
    K = np.exp(1.0 - 0.5 * np.exp(-0.7) * (t[:, None] - t[None, :]) ** 2) + 0.25 * np.eye(N)
    A = np.eye(Np)
    A[:N, :N] = K
    return K, torch.from_numpy(A.reshape(-1).copy()).cuda(), Np


def test_static_workspace_inverse_equals_caller_workspace_inverse():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    from magma_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ld = torch.zeros(2, dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    for k, N in enumerate([700, 300, 1024, 1]):  # second call smaller than the first
        K, A0, Np = _problem(torch, lib, N, 10 + k)
        a_ref, a_new = A0.clone(), A0.clone()
        work = torch.empty((lib.magma_workspace_bytes(N) + 7) // 8, dtype=torch.float64, device="cuda")
        _lib.call("magma_spd_inverse", N, _p(a_ref), _p(ld[0:]), _p(info[0:]), _p(work), st)
        torch.cuda.synchronize()
        # free device memory as the driver sees it (hipMemGetInfo): a hipMalloc inside the library
        # would show here, which torch's own allocator statistics cannot see. The first call on a
        # device may load the code object's static array; from the second call on nothing may move.
        before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        _lib.call("magma_spd_inverse_static", N, _p(a_new), _p(ld[1:]), _p(info[1:]), st)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        if k > 0:
            assert torch.cuda.mem_get_info()[0] == free_before, (N, free_before, torch.cuda.mem_get_info()[0])
        assert info.tolist() == [0, 0], (N, info.tolist())
        assert torch.equal(a_new, a_ref), (N, (a_new - a_ref).abs().max().item())
        assert ld[0].item() == ld[1].item()
        # and against an independent inverse, to the suite's tolerance for the dense inverse
        got = a_new.view(Np, Np)[:N, :N].cpu().numpy()
        exp = np.linalg.inv(K)
        assert np.abs(got - exp).max() < 1e-9 * np.abs(exp).max(), N
        assert abs(ld[1].item() - np.linalg.slogdet(K)[1]) < 1e-10 * max(1.0, abs(ld[1].item()))
    # an indefinite matrix is flagged the same way
    A = np.eye(128) + 0.01
    A[70, 70] = -1.0
    bad = torch.from_numpy(A.reshape(-1).copy()).cuda()
    _lib.call("magma_spd_inverse_static", 128, _p(bad), _p(ld[1:]), _p(info[1:]), st)
    torch.cuda.synchronize()
    assert int(info[1].item()) == 71
    # beyond the built-in workspace: refused with a clear message, nothing launched
    big = torch.zeros(8, dtype=torch.float64, device="cuda")
    rc = lib.magma_spd_inverse_static(1025, _p(big), _p(ld[1:]), _p(info[1:]), st)
    assert rc == 2 and b"1024" in lib.magma_last_error()
