"""The C-ABI library loads and exports every symbol include/hecka.h declares (no compute calls: no GPU)."""
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from hyper_amd import _lib
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'hecka.h')).read()
    declared = set(re.findall(r'\b(hecka_\w+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.hecka_version() >= 1


def test_bad_arguments_return_error_codes_without_a_gpu():
    from hyper_amd import _lib
    lib = _lib.lib()
    assert lib.hecka_colmean_f32(None, 1, 1, 1, None, None, None) == -1
    assert lib.hecka_pairwise_workspace_bytes(0, 1, 1) == 0
    assert lib.hecka_pairwise_workspace_bytes(64, 64, 512 * 512) > 0
    # the GEMM arithmetic is a per-call argument (no process-wide mode in the library): an unknown value is refused
    assert lib.hecka_gram_bwd_f32(None, None, None, None, None, 1, 1, 1, None, 7, None) == -1
    assert b'gemm_mode' in lib.hecka_last_error()
    assert not hasattr(lib, 'hecka_set_gemm_mode')


def test_gemm_mode_is_per_thread_and_per_node():
    """ops.set_gemm_mode = process default; `with ops.gemm_mode(m)` overrides it for the calling thread only."""
    import threading
    from hyper_amd import ops
    assert ops.get_gemm_mode() == 0
    seen = {}
    with ops.gemm_mode(2):
        assert ops.get_gemm_mode() == 2
        t = threading.Thread(target=lambda: seen.setdefault('other', ops.get_gemm_mode()))
        t.start()
        t.join()
        with ops.gemm_mode(1):
            assert ops.get_gemm_mode() == 1
        assert ops.get_gemm_mode() == 2
    assert seen['other'] == 0 and ops.get_gemm_mode() == 0
    try:
        ops.set_gemm_mode(1)
        assert ops.get_gemm_mode() == 1
    finally:
        ops.set_gemm_mode(0)


def test_cpu_tensors_are_rejected_not_emulated():
    import pytest
    import torch
    from hyper_amd import ops
    with pytest.raises(RuntimeError):
        ops.centered_gram_pairwise(torch.randn(2, 4, 3))
