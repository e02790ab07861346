"""The options of the functional API that no shipped config uses (VERDICT r01 "missing" 3-4): readd_diag=False,
Y != None Grams, log / hyperenergy / learned Grams, learnable / per-member / EMA rbf bandwidth, grad_var, EMA centring,
half_space, normalize=True.  tests/golden/functional_f64.npz holds the real reference's float64 outputs and gradients
(make_golden.py::run_functional).  CPU: the oracle restatement against it (1e-12).  GPU: hyper_amd (libhecka GEMMs + the
reference's element-wise formulas on the device, float32) against it, rel 1e-5 of the largest entry."""
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
GOLD = np.load(os.path.join(HERE, 'golden', 'functional_f64.npz'), allow_pickle=False)
M_N_L_P = (4, 7, 5, 6)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def t(name, dtype, dev, grad=True):
    return torch.from_numpy(GOLD[name]).to(dtype).to(dev).requires_grad_(grad)


def run(name, fn, dtype, dev, tol, has_y=False, extra=None):
    X = t(name + '/x', dtype, dev)
    Y = t(name + '/y', dtype, dev) if has_y else None
    out = fn(X, Y)
    assert tuple(out.shape) == GOLD[name + '/out'].shape, name
    assert rel(out.detach().cpu().numpy(), GOLD[name + '/out']) <= tol, (name, 'out')
    (out * t(name + '/up', dtype, dev, False)).sum().backward()
    assert rel(X.grad.cpu().numpy(), GOLD[name + '/dx']) <= tol, (name, 'dx')
    if has_y and np.abs(GOLD[name + '/dy']).max() > 0:
        assert rel(Y.grad.cpu().numpy(), GOLD[name + '/dy']) <= tol, (name, 'dy')
    if has_y and np.abs(GOLD[name + '/dy']).max() == 0:
        assert Y.grad is None or float(Y.grad.abs().max()) == 0.0
    for key, got in (extra() if extra else {}).items():
        assert rel(got.detach().cpu().numpy(), GOLD[name + '/' + key]) <= tol, (name, key)


def cases(api, dtype, dev, tol):
    """api: an object with the reference's function names (oracle adapter or hyper_amd)."""
    bg = api.batched_gram
    run('linear_readd_false', lambda x, y: bg(x, None, 'linear', True, False, True), dtype, dev, tol)
    run('linear_cross', lambda x, y: bg(x, y, 'linear', False, True, False), dtype, dev, tol, True)
    run('linear_cross_centered', lambda x, y: bg(x, y, 'linear', False, True, True), dtype, dev, tol, True)
    run('log', lambda x, y: bg(x, None, 'log', True, True, True), dtype, dev, tol)
    run('cossim_readd_false', lambda x, y: bg(x, None, 'cossim', True, False, True), dtype, dev, tol)
    run('cossim_cross', lambda x, y: bg(x, y, 'cossim', False, True, False), dtype, dev, tol, True)
    run('hyperenergy', lambda x, y: bg(x, None, 'hyperenergy', True, True, True, arc_eps=0.005, eps=1e-4, mhe_s=2), dtype, dev, tol)
    run('rbf_readd_false', lambda x, y: bg(x, None, 'rbf', True, False, True, param=1.7, median=True), dtype, dev, tol)
    run('rbf_cross_fixed', lambda x, y: bg(x, y, 'rbf', False, True, False, param=1.3, median=False), dtype, dev, tol, True)
    run('rbf_cross_median_detach_right', lambda x, y: bg(x, y, 'rbf', False, True, True, param=0.8, median=True, detach_right=True),
        dtype, dev, tol, True)
    for nm in ('rbf_learn_scalar', 'rbf_learn_members'):
        par = t(nm + '/param', dtype, dev)
        run(nm, lambda x, y: bg(x, None, 'rbf', True, True, True, param=par, median=False), dtype, dev, tol,
            extra=lambda: {'dparam': par.grad})
    Xg = t('rbf_grad_var/x', dtype, dev)
    g, dg = api.rbf_gram(Xg, None, param=1.2, median=True, detach_diag=True, readd=True, grad_var=Xg)
    assert rel(g.detach().cpu().numpy(), GOLD['rbf_grad_var/out']) <= tol and rel(dg.cpu().numpy(), GOLD['rbf_grad_var/grad']) <= tol
    ma = api.EMA(0.9)
    random.seed(3)      # the member whose median feeds the moving average is drawn with random.randint, as in the reference
    for i in range(3):
        run('rbf_ema_%d' % i, lambda x, y: bg(x, None, 'rbf', True, True, True, param=1.1, median=True, median_ma=ma, median_update=True),
            dtype, dev, tol, extra=lambda: {'shadow': ma.get()})
    run('rbf_ema_frozen', lambda x, y: bg(x, None, 'rbf', True, True, True, param=1.1, median=True, median_ma=ma, median_update=False),
        dtype, dev, tol)
    cma = api.EMA(0.8)
    for i in range(3):
        run('center_ema_%d' % i, lambda x, y: bg(x, None, 'linear', True, True, True, ma=cma, update_ma=True), dtype, dev, tol,
            extra=lambda: {'shadow': cma.get()})
    model = torch.nn.Linear(M_N_L_P[3], 5).to(dtype).to(dev)
    with torch.no_grad():
        model.weight.copy_(t('learned_linear/w', dtype, dev, False))
        model.bias.copy_(t('learned_linear/b', dtype, dev, False))
    run('learned_linear', lambda x, y: bg(x, None, 'learned', True, True, True, model=model, learned_kernel='linear'), dtype, dev, tol,
        extra=lambda: {'dw': model.weight.grad, 'db': model.bias.grad})
    for nm, kw in (('half_space', {}), ('half_space_detach_right', {'detach_right': True}),
                   ('half_space_exp_all', {'use_exp': True, 'remove_diag': False})):
        run(nm, lambda x, y: api.hyperspherical_energy(x, s=2.0, half_space=True, eps=0.0025, arc_eps=0.031, **kw), dtype, dev, tol)
    run('he_normalize', lambda x, y: api.he_normalized(x, tuple(GOLD['he_normalize/minmax'])), dtype, dev, tol)


class OracleAPI(object):
    """The oracle's functions under the reference's call signatures."""

    def __init__(self):
        from oracle import functional_ref as F, hecka_ref as R
        self.F, self.R, self.EMA = F, R, F.EMA

    def rbf_gram(self, X, Y=None, grad_var=None, **kw):
        g = self.F.rbf_gram(X, Y, **kw)
        return (g, torch.autograd.grad(g.sum(), grad_var, retain_graph=True)[0]) if grad_var is not None else g

    def batched_gram(self, X, Y, kernel, detach_diag, readd, center, ma=None, update_ma=True, **kw):
        F = self.F
        if kernel == 'learned':
            B, N, P = X.shape
            X = kw['model'](X.reshape(B * N, P)).reshape(B, N, -1)
            kernel = kw['learned_kernel']
            kw = {}
        if kernel == 'hyperenergy':
            gram = F.hyperenergy_gram(X, Y, **kw)
        elif kernel == 'rbf':
            gram = F.rbf_gram(X, Y, detach_diag=detach_diag, readd=readd, **kw)
        else:
            gram = {'linear': F.linear_gram, 'log': F.log_gram, 'cossim': F.cossim_gram}[kernel](X, Y, detach_diag, readd)
        return F.center_gram(gram, ma, update_ma) if center else gram

    def hyperspherical_energy(self, x, s, half_space, eps, arc_eps, **kw):
        return self.F.half_space_energy(x, s, eps, arc_eps, **kw)

    def he_normalized(self, feats, minmax):
        e = self.R.he_layer(feats, he_s=2.0, eps=float(np.float32(0.0025)), arc_eps=float(np.float32(0.031)), detach_diag=True)
        return self.F.normalize_energy(e, *minmax)


class HipAPI(object):
    def __init__(self):
        from hyper_amd.diversity import kernels as K, hyperspherical_energy
        self.K, self.EMA, self.rbf_gram, self.batched_gram = K, (lambda d: K.EMA(d).train()), K.rbf_gram, K.batched_gram
        self.hyperspherical_energy = hyperspherical_energy

    def he_normalized(self, feats, minmax):
        from hyper_amd.diversity import model_kernels as mk
        kern = mk.build_model_kernel({'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031, 'normalize': True,
                                      'feature_kernel': {'name': 'feature', 'kernel': 'linear', 'detach_diag': True}}).to(feats.device)
        arc, eps = kern.get_smoothing()
        mk.HE_VALUE_CACHE[(feats.shape[0], feats.shape[1], arc.item(), eps.item())] = minmax
        return kern(None, feats, None, {})


def test_oracle_restatement_matches_the_reference_fixture():
    cases(OracleAPI(), torch.float64, torch.device('cpu'), 1e-12)


@pytest.mark.gpu
def test_hip_general_path_matches_the_reference_fixture():
    cases(HipAPI(), torch.float32, torch.device('cuda:0'), 1e-5)


@pytest.mark.gpu
def test_normalize_presolve_runs_and_brackets_random_energies():
    """normalize=True without a cached entry: the 100-step pre-solve (model_kernels.py:302-367) runs on the device, caches
    (min, max) under the reference's key and maps the energies affinely; min < max, and the normalised energies of random
    features lie between the extremes."""
    from hyper_amd.diversity import model_kernels as mk
    dev = torch.device('cuda:0')
    mk.HE_VALUE_CACHE.clear()
    cfg = {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031, 'normalize': True,
           'feature_kernel': {'name': 'feature', 'kernel': 'linear', 'detach_diag': True}}
    kern = mk.build_model_kernel(cfg).to(dev)
    torch.manual_seed(0)
    feats = torch.randn(6, 10, 24, device=dev, requires_grad=True)
    out = kern(None, feats, None, {})
    assert len(mk.HE_VALUE_CACHE) == 1
    (lo, hi), = mk.HE_VALUE_CACHE.values()
    assert np.isfinite(lo) and np.isfinite(hi) and lo < hi
    assert float(out.min()) > -0.05 and float(out.max()) < 1.0
    out.mean().backward()
    plain = mk.build_model_kernel(dict(cfg, normalize=False)).to(dev)(None, feats.detach(), None, {})
    assert rel(out.detach().cpu().numpy(), ((plain - lo) / (hi - lo)).cpu().numpy()) <= 1e-5
    assert torch.isfinite(feats.grad).all()
