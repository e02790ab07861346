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
    assert lib.hecka_set_gemm_mode(7) == -1 and lib.hecka_get_gemm_mode() == 0
    assert lib.hecka_set_gemm_mode(2) == 0 and lib.hecka_get_gemm_mode() == 2
    assert lib.hecka_set_gemm_mode(1) == 0 and lib.hecka_get_gemm_mode() == 1
    assert lib.hecka_set_gemm_mode(0) == 0


def test_cpu_tensors_are_rejected_not_emulated():
    import pytest
    import torch
    from hyper_amd import ops
    with pytest.raises(RuntimeError):
        ops.centered_gram_pairwise(torch.randn(2, 4, 3))
