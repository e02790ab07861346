"""hecka_pairwise_noalloc_f32 - the entry point for callers that cannot allocate (include/hecka.h) - through the C-ABI
against hecka_pairwise_f32, which takes the same partial-sum buffer from the caller: same kernels, same order, so the
results must be IDENTICAL (and both within 5e-6 of fp64).  Calls in a row, each with a smaller input than the first: the
built-in scratch keeps the previous call's contents and nothing may depend on them.

"Allocates nothing" is checked at the DRIVER level: torch's own counters only see torch's caching allocator, a hipMalloc
inside libhecka would move `torch.cuda.mem_get_info()` (hipMemGetInfo: free bytes of the device) and nothing else.  The
set-up call (hecka_noalloc_init: hipGetSymbolAddress, code-object load) is made first and Python's cyclic GC is held off
around the measured calls, so that tensors of earlier tests being collected cannot move either figure."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(L, C, S, ws, nbytes, st):
    Ma, K = C.shape[0], C[0].numel()
    assert L.hecka_pairwise_f32(C.data_ptr(), Ma, C.data_ptr(), Ma, K, S.data_ptr(), ws.data_ptr(), nbytes, st) == 0


class _NoAllocation:
    """Context: the device's free memory and torch's allocation counter are the same on exit as on entry."""

    def __enter__(self):
        gc.collect()
        gc.disable()
        torch.cuda.synchronize()
        self.free0 = torch.cuda.mem_get_info()[0]
        self.count0 = torch.cuda.memory_stats().get('allocation.all.allocated', 0)
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                torch.cuda.synchronize()
                assert torch.cuda.mem_get_info()[0] == self.free0, 'device free memory changed across the call'
                assert torch.cuda.memory_stats().get('allocation.all.allocated', 0) == self.count0
        finally:
            gc.enable()
        return False


def test_pairwise_noalloc_matches_workspace_entry_point():
    from hyper_amd import _lib
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    assert L.hecka_noalloc_init() == 0, L.hecka_last_error()
    assert L.hecka_noalloc_init() == 0          # idempotent
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
        # one unmeasured call per shape (the HIP runtime may grow its own launch bookkeeping at a kernel's first use),
        # then the measured one: a hipMalloc made by the library per call would still show in it
        assert L.hecka_pairwise_noalloc_f32(C.data_ptr(), M, C.data_ptr(), M, K, S.data_ptr(), st) == 0
        S.fill_(float('nan'))
        with _NoAllocation():
            rc = L.hecka_pairwise_noalloc_f32(C.data_ptr(), M, C.data_ptr(), M, K, S.data_ptr(), st)
            assert rc == 0, L.hecka_last_error()
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
    with _NoAllocation():
        assert L.hecka_pairwise_noalloc_f32(Ca.data_ptr(), 8, Cb.data_ptr(), 24, 4096, S.data_ptr(), st) == 0
    assert torch.equal(S, S_ref)


def test_pairwise_noalloc_under_graph_capture_after_init():
    """The caller this entry point exists for: set up outside the capture, then captured and replayed."""
    from hyper_amd import _lib
    L = _lib.lib()
    dev = torch.device('cuda:0')
    assert L.hecka_noalloc_init() == 0, L.hecka_last_error()
    g = torch.Generator().manual_seed(12)
    M, B = 16, 64
    C = torch.randn(M, B, B, generator=g).to(dev)
    K = B * B
    nbytes = L.hecka_pairwise_workspace_bytes(M, M, K)
    ws = torch.empty((nbytes // 4,), device=dev)
    S_ref = torch.empty((M, M), device=dev)
    _pair(L, C, S_ref, ws, nbytes, torch.cuda.current_stream().cuda_stream)
    S = torch.zeros((M, M), device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = torch.cuda.current_stream().cuda_stream
        rc = L.hecka_pairwise_noalloc_f32(C.data_ptr(), M, C.data_ptr(), M, K, S.data_ptr(), st)
    assert rc == 0, L.hecka_last_error()
    for _ in range(2):
        S.zero_()
        graph.replay()
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
