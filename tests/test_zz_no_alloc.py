"""hecka_pairwise_noalloc_f32 - the entry point for callers that cannot allocate (include/hecka.h) - through the C-ABI
against hecka_pairwise_f32, which takes the same partial-sum buffer from the caller: same kernels, same order, so the
results must be IDENTICAL (and both within 5e-6 of fp64).  Two calls in a row, the second with a smaller input: the
built-in scratch keeps the first call's contents and nothing may depend on them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(L, C, S, ws, nbytes, st):
    Ma, K = C.shape[0], C[0].numel()
    assert L.hecka_pairwise_f32(C.data_ptr(), Ma, C.data_ptr(), Ma, K, S.data_ptr(), ws.data_ptr(), nbytes, st) == 0


def test_pairwise_noalloc_matches_workspace_entry_point():
    from hyper_amd import _lib
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    # bench shape (M = 64, B = 512: the streaming kernel), then smaller inputs on the LDS-tiled kernel (ragged M and K)
    for M, B in ((64, 512), (5, 300), (64, 128), (3, 37)):
        C = torch.randn(M, B, B, generator=g).to(dev)
        K = B * B
        nbytes = L.hecka_pairwise_workspace_bytes(M, M, K)
        assert nbytes <= 8 << 20
        ws = torch.empty((max(nbytes, 4) // 4,), device=dev)
        S_ref = torch.empty((M, M), device=dev)
        _pair(L, C, S_ref, ws, nbytes, st)
        S = torch.full((M, M), float('nan'), device=dev)
        mem0 = torch.cuda.memory_allocated()
        rc = L.hecka_pairwise_noalloc_f32(C.data_ptr(), M, C.data_ptr(), M, K, S.data_ptr(), st)
        assert rc == 0, L.hecka_last_error()
        assert torch.cuda.memory_allocated() == mem0
        torch.cuda.synchronize()
        assert torch.equal(S, S_ref), (M, B)
        Cd = C.double().reshape(M, -1)
        exact = Cd @ Cd.t()
        assert float((S.double() - exact).abs().max() / exact.abs().max()) <= 5e-6
    # rectangular (local members against all members, the sharded `gather` scheme)
    Ca, Cb = torch.randn(8, 64 * 64, generator=g).to(dev), torch.randn(24, 64 * 64, generator=g).to(dev)
    nbytes = L.hecka_pairwise_workspace_bytes(8, 24, 4096)
    ws = torch.empty((nbytes // 4,), device=dev)
    S_ref, S = torch.empty((8, 24), device=dev), torch.empty((8, 24), device=dev)
    assert L.hecka_pairwise_f32(Ca.data_ptr(), 8, Cb.data_ptr(), 24, 4096, S_ref.data_ptr(), ws.data_ptr(), nbytes, st) == 0
    assert L.hecka_pairwise_noalloc_f32(Ca.data_ptr(), 8, Cb.data_ptr(), 24, 4096, S.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(S, S_ref)


def test_pairwise_noalloc_rejects_what_does_not_fit():
    from hyper_amd import _lib
    L = _lib.lib()
    dev = torch.device('cuda:0')
    M, K = 2048, 1 << 16      # one slab of [2048, 2048] partial sums = 16 MiB > 8 MiB
    assert L.hecka_pairwise_workspace_bytes(M, M, K) > 8 << 20
    C = torch.zeros((M, 8), device=dev)      # never read: the call must fail before any launch
    S = torch.zeros((M, M), device=dev)
    rc = L.hecka_pairwise_noalloc_f32(C.data_ptr(), M, C.data_ptr(), M, K, S.data_ptr(), None)
    assert rc == -3
    assert b'built-in scratch' in L.hecka_last_error()
    assert L.hecka_pairwise_noalloc_f32(None, 4, None, 4, 16, S.data_ptr(), None) == -1
