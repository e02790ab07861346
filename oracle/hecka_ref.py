"""CPU oracle for the HE-CKA diversity-loss hot path.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement (stock PyTorch, dtype
generic, autograd for the backward) of the reference algorithm.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it;
the product path (`hyper_amd`) never does and fails loudly without its HIP library.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference
from /root/reference in the build container and stores its outputs under
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this restatement against
those fixtures (the reference itself has no tests for this path).

Every function cites the reference file:line whose behaviour it restates.  Shapes
use the survey's notation: M members, B samples, P features.
"""
import math
from collections import OrderedDict

import torch


# --------------------------------------------------------------------------- Grams
def detach_diagonal(gram, readd=True):
    """hyper/diversity/kernels.py:76-104 — diagonal keeps its value (if readd) but
    receives no gradient; off-diagonal untouched."""
    n = gram.shape[1]
    off = 1.0 - torch.eye(n, dtype=gram.dtype, device=gram.device)
    out = gram * off
    if readd:
        out = out + torch.diag_embed(torch.diagonal(gram, dim1=1, dim2=2)).detach()
    return out


def linear_gram(X, detach_diag=True, readd=True):
    """hyper/diversity/kernels.py:107-127 — K_m = X_m X_m^T."""
    gram = torch.bmm(X, X.transpose(1, 2))
    return detach_diagonal(gram, readd) if detach_diag else gram


def cossim_gram(X, detach_diag=True, readd=True):
    """hyper/diversity/kernels.py:145-169 — K_ab = x_a.x_b / sqrt(|x_a|^2 |x_b|^2);
    no eps (a zero row yields NaN, as in the reference)."""
    gram = torch.bmm(X, X.transpose(1, 2))
    sq = X.square().sum(dim=2, keepdim=True)
    gram = gram / torch.sqrt(torch.bmm(sq, sq.transpose(1, 2)))
    return detach_diagonal(gram, readd) if detach_diag else gram


def rbf_gram(X, param=1.0, median=True, detach_diag=True, readd=True):
    """hyper/diversity/kernels.py:198-286 (Y=None, no EMA) — squared distances from
    the Gram, diagonal detached, per-member bandwidth h_m = param * median(d_m) /
    (2 log(B+1)) taken without gradient (torch.median = lower median of the B^2
    values), K = exp(-d / (2 h_m)); median=False uses h = param^2."""
    G = torch.bmm(X, X.transpose(1, 2))
    dg = torch.diagonal(G, dim1=1, dim2=2)
    d = -2.0 * G + dg.unsqueeze(2) + dg.unsqueeze(1)
    if detach_diag:
        d = detach_diagonal(d, readd)
    M = X.shape[0]
    if median:
        with torch.no_grad():
            w = 2.0 * math.log(X.shape[1] + 1)
            h = torch.stack([float(param) * d[m].median() / w for m in range(M)]).view(M, 1, 1)
    else:
        h = torch.full((M, 1, 1), float(param) ** 2, dtype=X.dtype, device=X.device)
    return torch.exp(-d / (2.0 * h))


GRAMS = {'linear': linear_gram, 'cossim': cossim_gram, 'rbf': rbf_gram}


def center_gram(gram):
    """hyper/diversity/kernels.py:335-345 — ONE-sided: subtract each row's mean."""
    return gram - gram.mean(dim=2, keepdim=True)


def feature_gram(feats, kernel='linear', detach_diag=True, readd=True, **params):
    """hyper/diversity/model_kernels.py:84-123 — flatten [M,B,*] to [M,B,P], build the
    (uncentered) Gram."""
    if feats.ndim < 3:
        raise RuntimeError('features need at least 3 dims (members, samples, features)')
    X = feats.reshape(feats.shape[0], feats.shape[1], -1)
    return GRAMS[kernel](X, detach_diag=detach_diag, readd=readd, **params)


# ------------------------------------------------------------------- pairwise heads
def unit_vector_rows(grams, eps):
    """hyper/diversity/__init__.py:6-30 — vec(C_m)/(||vec(C_m)||_2 + eps)."""
    v = grams.reshape(grams.shape[0], -1) if grams.ndim == 3 else grams
    if v.ndim != 2:
        raise ValueError('Expected matrix of [B, P]')
    return v / (torch.linalg.vector_norm(v, ord=2, dim=1, keepdim=True) + eps)


def _offdiag(mat):
    return mat[~torch.eye(*mat.shape, dtype=torch.bool, device=mat.device)]


def pairwise_cossim(grams, eps=1e-6, reduction='off_diag', detach_right=False):
    """hyper/diversity/__init__.py:33-56 — CKA head."""
    u = unit_vector_rows(grams, eps)
    cos = u @ (u.t().detach() if detach_right else u.t())
    return _offdiag(cos) if reduction == 'off_diag' else cos


def hyperspherical_energy(grams, s=0.0, eps=3e-5, arc_eps=3e-4, remove_diag=True,
                          detach_right=False, abs_vals=False, use_exp=False):
    """hyper/diversity/__init__.py:79-155 (half_space=False, the only way it is called)."""
    u = unit_vector_rows(grams, eps)
    cos = u @ (u.t().detach() if detach_right else u.t())
    if abs_vals:
        cos = cos.abs()
    if remove_diag:
        cos = _offdiag(cos)
    theta = torch.arccos(cos / (1.0 + arc_eps))
    if s == 0.0 or s == 1.0:
        energy = 1.0 / (theta + eps)
        if s == 0.0:
            energy = torch.log(energy + eps)
    elif use_exp:
        energy = torch.exp(-(s * theta + eps))
    else:
        energy = (1.0 + eps) / (theta ** s + eps)
    return energy


# ------------------------------------------------------------------- model kernels
def he_layer(feats, he_s=2.0, eps=1e-6, arc_eps=1e-3, center=True, reduction='off_diag',
             abs_vals=False, detach_right=False, use_exp=False, kernel='linear',
             detach_diag=False, readd=True, kparams=None, track=None):
    """hyper/diversity/model_kernels.py:369-407 (normalize=False) — returns the energy
    vector [M(M-1)] (off_diag) or matrix [M,M]; eps/arc_eps enter by absolute value."""
    gram = feature_gram(feats, kernel, detach_diag, readd, **(kparams or {}))
    if center:
        gram = center_gram(gram)
    if track is not None:
        track['gram'] = gram
    return hyperspherical_energy(gram, s=he_s, eps=abs(eps), arc_eps=abs(arc_eps),
                                 remove_diag=(reduction == 'off_diag'), detach_right=detach_right,
                                 abs_vals=abs_vals, use_exp=use_exp)


def cka_layer(feats, eps=1e-6, center=True, reduction='off_diag', detach_right=False,
              kernel='linear', detach_diag=False, readd=True, kparams=None, track=None):
    """hyper/diversity/model_kernels.py:231-249 — abs_vals is stored but never applied."""
    gram = feature_gram(feats, kernel, detach_diag, readd, **(kparams or {}))
    if center:
        gram = center_gram(gram)
    if track is not None:
        track['gram'] = gram
    return pairwise_cossim(gram, eps=eps, reduction=reduction, detach_right=detach_right)


def layer_weights(name, total, **cfg):
    """hyper/diversity/layer_weighting.py — host-side per-layer scalars for the
    non-learnable schemes: list :144-172, linear :175-196, uniform :199-206,
    exponential :209-230, start_end_linear :233-266."""
    o = cfg.get('ord', None)

    def apply_ord(v):
        if o is None:
            return v
        if o == 1:
            return abs(v)
        if o >= 2:
            return abs(v) ** o
        raise ValueError

    def raw(i):
        if name == 'list':
            if i >= len(cfg['weights']):
                raise ValueError('Layer {} not in weights'.format(i))
            return apply_ord(cfg['weights'][i])
        if name == 'uniform':
            return apply_ord(1.0)
        if name == 'linear':
            return apply_ord(cfg.get('start_weight', 1.0) + cfg.get('weight_increase', 1.0) * i)
        if name == 'exponential':
            return apply_ord(cfg.get('start_weight', 1.0) * cfg.get('weight_increase', 1.0) ** i)
        if name == 'start_end_linear':
            if i == 0:
                v = cfg['first_layer']
            elif i == total - 1 and cfg.get('last_layer') is not None:
                v = cfg['last_layer']
            else:
                v = cfg['middle_start'] + cfg['middle_increase'] * (i - 1)
            return apply_ord(v)
        raise KeyError(name)

    vals = [raw(i) for i in range(total)]
    if name == 'list' and not cfg.get('normalize', True):
        return vals
    tot = sum(vals)
    if o is not None and o == 2:
        tot = math.sqrt(tot)
    elif o is not None and o > 2:
        tot = tot ** (1.0 / o)
    return [v / tot for v in vals]


def model_layers(feats, layer_fn, weights, reduction='mean', track=None):
    """hyper/diversity/model_kernels.py:446-478 — sum_l w_l * reduce(layer kernel_l)."""
    if len(feats) == 0:
        return None
    layer_track = [OrderedDict() for _ in feats]
    total = None
    for i, f in enumerate(feats.values() if isinstance(feats, dict) else feats):
        k = layer_fn(f, track=layer_track[i])
        if reduction == 'mean':
            k = k.mean()
        elif reduction == 'sum':
            k = k.sum()
        total = weights[i] * k if total is None else total + weights[i] * k
    if track is not None:
        track['layer_track'] = layer_track
    return total


def loss_and_grads(feats, layer_cfg, weights, reduction='mean', head='he'):
    """One "eval" of the benchmark metric: scalar loss + dLoss/dX for every layer,
    through autograd exactly as the reference trainer does (methods.py:139-152)."""
    xs = [f.detach().clone().requires_grad_(True) for f in feats]
    fn = he_layer if head == 'he' else cka_layer
    loss = model_layers(xs, lambda f, track: fn(f, track=track, **layer_cfg), weights, reduction)
    grads = torch.autograd.grad(loss, xs)
    return loss.detach(), [g.detach() for g in grads]


def f32(v):
    """Value of a python float after storage in an fp32 tensor (torch.tensor(float(v)))."""
    return float(torch.tensor(float(v), dtype=torch.float32))


def run_config(cfg, feats, track=None):
    """Evaluate a YAML-shaped `model_layers` config dict (the reference's own schema,
    model_kernels.py:480-491) with this restatement.  feats: list/OrderedDict of [M,B,*]."""
    assert cfg['name'] == 'model_layers'
    lk = cfg['layer_kernel']
    assert not isinstance(lk, list), 'per-layer kernel lists are evaluated by the caller'
    fk = lk['feature_kernel']
    kw = {k: v for k, v in lk.items() if k not in ('name', 'feature_kernel', 'learnable', 'normalize')}
    # the reference stores eps / arc_eps / kernel params as fp32 0-dim Parameters
    # (model_kernels.py:81,229,284-285), so their values are fp32-rounded whatever the feature dtype
    for k in ('eps', 'arc_eps'):
        if k in kw:
            kw[k] = f32(kw[k])
    kw.update(kernel=fk['kernel'], detach_diag=fk.get('detach_diag', False), readd=fk.get('readd_diag', True))
    kp = dict(fk.get('params') or {})
    if fk['kernel'] == 'rbf':
        kp.setdefault('param', 1.0)
    kw['kparams'] = {k: f32(v) for k, v in kp.items()}
    wcfg = dict(cfg['layer_weighting'])
    weights = layer_weights(wcfg.pop('name'), len(feats), **wcfg)
    fn = he_layer if lk['name'] == 'he' else cka_layer
    return model_layers(feats, lambda f, track: fn(f, track=track, **kw), weights,
                        cfg.get('reduction', 'mean'), track)


# --------------------------------------------------------------------------- evaluation helper
def unbiased_hsic(K, L):
    """hyper/generators/base.py:210-221 `_uHSIC` - unbiased HSIC of two zero-diagonal Gram matrices [N,N]."""
    N = K.shape[0]
    ones = torch.ones(N, 1, dtype=K.dtype)
    result = torch.trace(K @ L)
    result = result + ((ones.t() @ K @ ones @ ones.t() @ L @ ones) / ((N - 1) * (N - 2))).item()
    result = result - ((ones.t() @ K @ L @ ones) * 2 / (N - 2)).item()
    return (1 / (N * (N - 3)) * result).item()


def unbiased_cka(batches):
    """hyper/generators/base.py:239-313 - `batches`: per batch a list of L feature tensors [M,B,P_l].  Member pairs
    i < j, layer pairs with layer_j <= layer_i (K from member i at layer_i, L from member j at layer_j), mirrored into
    [layer_j, layer_i]; HSIC terms summed over batches (each divided by the number of batches), CKA of the sums, mean
    over the member pairs -> [L,L] float64."""
    num_train = len(batches)
    hs = None
    for features in batches:
        mbs, nl = features[0].shape[0], len(features)
        if hs is None:
            hs = torch.zeros((mbs, mbs, nl, nl, 3), dtype=torch.double)
        for m_i in range(mbs):
            for m_j in range(m_i + 1, mbs):
                for layer_i in range(nl):
                    f1 = features[layer_i][m_i]
                    K = f1 @ f1.t()
                    K.fill_diagonal_(0.0)
                    for layer_j in range(layer_i + 1):
                        f2 = features[layer_j][m_j]
                        Lm = f2 @ f2.t()
                        Lm.fill_diagonal_(0.0)
                        vals = (unbiased_hsic(K, K) / num_train, unbiased_hsic(K, Lm) / num_train,
                                unbiased_hsic(Lm, Lm) / num_train)
                        for t, v in enumerate(vals):
                            hs[m_i, m_j, layer_i, layer_j, t] += v
                            if layer_i != layer_j:
                                hs[m_i, m_j, layer_j, layer_i, t] += v
    cka = (hs[..., 1] + 1e-8) / (torch.sqrt(torch.abs(hs[..., 0])) * torch.sqrt(torch.abs(hs[..., 2])) + 1e-8)
    avg, count = torch.zeros(cka.shape[2:], dtype=torch.double), 0
    for i in range(cka.shape[0]):
        for j in range(i + 1, cka.shape[1]):
            avg += cka[i, j]
            count += 1
    return avg / count
