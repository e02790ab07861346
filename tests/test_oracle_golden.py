"""The CPU oracle (oracle/hecka_ref.py) against the golden vectors produced by the real
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from oracle import hecka_ref as R


def _run(rec, dtype):
    xs = [torch.from_numpy(rec['x%d' % i]).to(dtype).requires_grad_(True) for i in range(rec['L'])]
    track = {}
    out = R.run_config(rec['config'], xs, track)
    if out.ndim == 0:
        out.backward()
    else:
        (out * torch.from_numpy(rec['upstream']).to(dtype)).sum().backward()
    return out.detach().numpy(), [x.grad.numpy() for x in xs], [t['gram'].detach().numpy() for t in track['layer_track']]


@pytest.mark.parametrize('name', golden_cases('f64'))
def test_oracle_matches_reference_f64(name):
    rec = load_golden(name, 'f64')
    out, grads, grams = _run(rec, torch.float64)
    np.testing.assert_allclose(out, rec['out'], rtol=1e-12, atol=1e-14)
    for i, g in enumerate(grads):
        scale = np.abs(rec['dx%d' % i]).max()
        np.testing.assert_allclose(g, rec['dx%d' % i], rtol=0, atol=1e-9 * max(scale, 1e-300))
        if 'gram%d' % i in rec:
            np.testing.assert_allclose(grams[i], rec['gram%d' % i], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('name', golden_cases('f32'))
def test_oracle_matches_reference_f32(name):
    # same op sequence in fp32: identical up to the last few ulps of the fp32 pipeline
    rec = load_golden(name, 'f32')
    out, grads, _ = _run(rec, torch.float32)
    # he_lin_none keeps the diagonal pair (i,i): its gradient coefficient is eps/d^3, which
    # fp32 autograd forms as the cancelling difference 1/d^2 - n/d^3 -> noisy to ~1e-2 of the gradient scale
    tol = 2e-2 if name == 'he_lin_none' else 2e-4
    np.testing.assert_allclose(out, rec['out'], rtol=2e-4, atol=1e-6)
    for i, g in enumerate(grads):
        scale = np.abs(rec['dx%d' % i]).max()
        np.testing.assert_allclose(g, rec['dx%d' % i], rtol=0, atol=tol * max(scale, 1e-30))


def test_closed_form_matches_autograd():
    """The staged algebra the HIP kernels implement == reference autograd (fp64)."""
    from oracle import closed_form as CF
    torch.manual_seed(3)
    feats = [torch.randn(6, 16, p, dtype=torch.float64) for p in (5, 24, 40)]
    w = R.layer_weights('uniform', 3)
    for cfg in (dict(he_s=2.0, eps=0.0025, arc_eps=0.031), dict(he_s=2.0, eps=1e-3, arc_eps=1e-3, use_exp=True),
                dict(he_s=0.0, eps=1e-3, arc_eps=1e-2), dict(he_s=1.0, eps=1e-3, arc_eps=1e-2),
                dict(he_s=3.0, eps=1e-5, arc_eps=3e-3, abs_vals=True), dict(he_s=2.0, eps=1e-3, arc_eps=1e-2, detach_right=True)):
        l, g = R.loss_and_grads(feats, dict(kernel='linear', detach_diag=True, **cfg), w)
        l2, g2 = CF.he_loss_and_grad([f.numpy() for f in feats], w, s=cfg['he_s'], eps=cfg['eps'], arc_eps=cfg['arc_eps'],
                                     use_exp=cfg.get('use_exp', False), abs_vals=cfg.get('abs_vals', False),
                                     detach_right=cfg.get('detach_right', False))
        assert abs(l.item() - l2) <= 1e-12 * abs(l2)
        for a, b in zip(g, g2):
            assert np.abs(a.numpy() - b).max() <= 1e-11 * np.abs(b).max()


def test_unbiased_hsic_matches_reference_golden():
    """oracle.unbiased_hsic against `ModelGenerator._uHSIC` of the real reference (tests/golden/uhsic_f32.npz)."""
    import numpy as np
    import torch
    from oracle import hecka_ref as R
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'uhsic_f32.npz'))
    for idx in range(3):
        K, L = torch.from_numpy(z['K%d' % idx]), torch.from_numpy(z['L%d' % idx])
        got = [R.unbiased_hsic(K, K), R.unbiased_hsic(K, L), R.unbiased_hsic(L, L)]
        np.testing.assert_allclose(got, z['v%d' % idx], rtol=1e-6)
        got64 = [R.unbiased_hsic(K.double(), K.double()), R.unbiased_hsic(K.double(), L.double())]
        np.testing.assert_allclose(got64, z['v%d' % idx][:2], rtol=1e-4)
