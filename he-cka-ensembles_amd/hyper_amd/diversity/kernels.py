"""Batched Gram builders — functional API of the reference's hyper/diversity/kernels.py
(EMA :32, detach_diagonal :76, linear_gram :107, log_gram :130, cossim_gram :145, hyperenergy_gram :172, rbf_gram :197,
center_gram :335, learned_gram :374, batched_gram :425, vectorize_grams :456, KERN_PARAMS :13, MAPS :364), backed by
libhecka.

This is the UN-FUSED entry: each function is its own differentiable op, so Gram, centring and the
pairwise head each make a pass over [M,B,B].  The model kernels (model_kernels.py) use the fused
path (ops.centered_gram_pairwise) instead whenever the configuration allows it.

`linear`, `cossim`, `rbf` with the options the shipped YAMLs use run as single fused kernels.  The options no config
uses - Y != None, readd_diag=False, a learnable / EMA / per-member rbf bandwidth, grad_var, `log`, `hyperenergy`,
`learned` - take the GENERAL path below: the O(M N L P) products come from libhecka's grouped GEMM (differentiable in
both operands), the element-wise rest of the formula is written with torch ops on the [M,N,L] result exactly as the
reference writes it, on the GPU (pinned by tests/golden/functional_f64.npz, made by the real reference).  `laplacian`
raises in the reference itself (:301) and here."""
import math
import random

import torch
import torch.nn as nn

from .. import ops

KERN_PARAMS = {
    'linear': {},
    'log': {},
    'cossim': {'eps': 1e-6},
    'hyperenergy': {'arc_eps': 0.005, 'eps': 1e-6, 'mhe_s': 2},
    'rbf': {'param': 1.0},
    'laplacian': {'param': 1.0},
}


class EMA(nn.Module):
    """Exponential moving average of a tensor (kernels.py:32-73): the first value seen in training initialises the
    shadow, later ones move it by (1 - decay); `forward` returns the shadow."""

    def __init__(self, decay, initial=None):
        super().__init__()
        self.decay = decay
        self.ten = initial
        self.register_buffer('shadow', None if initial is None else initial.clone().detach(), persistent=True)

    @torch.no_grad()
    def update(self):
        if not self.training:
            return
        self.shadow.sub_((1.0 - self.decay) * (self.shadow - self.ten))

    def get(self):
        return self.shadow

    @torch.no_grad()
    def forward(self, inputs, update=True):
        if self.training and update:
            first = self.ten is None
            self.ten = inputs
            if first:
                self.shadow = inputs.clone().detach()
            else:
                self.update()
        return self.shadow


def _same(Y, X):
    return Y is None or Y is X


def _xy(X, Y):
    """X Y^T per member, differentiable in both operands (Y None: X X^T), nothing masked."""
    return ops.linear_gram(X, False) if _same(Y, X) else ops.cross_gram(X, Y)


def linear_gram(X, Y=None, detach_diag=True, readd=True, *args, **kwargs):
    """K_m = X_m Y_m^T; with detach_diag the diagonal gets no gradient and keeps its value (readd) or becomes zero."""
    if _same(Y, X) and not (detach_diag and not readd):
        return ops.linear_gram(X, detach_diag)
    K = _xy(X, Y)
    return detach_diagonal(K, readd) if detach_diag else K


def log_gram(X, Y=None, detach_diag=True, readd=True, *args, **kwargs):
    """log(|linear_gram| + 1e-6) (kernels.py:130-142)."""
    return torch.log(torch.abs(linear_gram(X, Y, detach_diag, readd, *args, **kwargs)) + 1e-6)


def _scalar(v):
    return float(v.detach()) if isinstance(v, torch.Tensor) else float(v)


def cossim_gram(X, Y=None, detach_diag=True, readd=True, *args, **kwargs):
    """K_ab = x_a.y_b / sqrt(|x_a|^2 |y_b|^2).  With Y = X the diagonal is identically 1 and carries no gradient, so
    detach_diag only matters together with readd=False.  (`eps` is accepted and ignored, as in the reference.)"""
    if _same(Y, X) and not (detach_diag and not readd):
        return ops.transformed_gram_ad(X, ops.GramSpec(ops.GRAM_COSSIM))
    Yv = X if Y is None else Y
    gram = _xy(X, Y)
    norms = torch.sqrt(torch.bmm(X.square().sum(dim=2, keepdim=True), Yv.square().sum(dim=2, keepdim=True).transpose(1, 2)))
    gram = gram / norms
    return detach_diagonal(gram, readd) if detach_diag else gram


def hyperenergy_gram(X, Y=None, detach_diag=True, readd=True, *args, **kwargs):
    """(1 + eps) / (arccos(cossim / (1 + arc_eps))^mhe_s + eps), element-wise on the cosine Gram (kernels.py:172-194)."""
    cosim = cossim_gram(X, Y, detach_diag=False, readd=True)
    arc_eps = 1.0 + kwargs.get('arc_eps', 0.005)
    eps = kwargs.get('eps', 1e-4)
    mhe_s = kwargs.get('mhe_s', 2)
    if not isinstance(arc_eps, float) and arc_eps.ndim > 0:
        arc_eps = arc_eps.view(arc_eps.shape[0], 1, 1)
        eps = eps.view(arc_eps.shape[0], 1, 1)
    arcs = torch.arccos(cosim / arc_eps)
    return (1.0 + eps) / (torch.pow(arcs, mhe_s) + eps)


def rbf_gram(X, Y=None, param=1.0, median=True, detach_diag=True, readd=True, grad_var=None, median_ma=None,
             median_update=True, detach_right=False, *args, **kwargs):
    """K = exp(-d / (2 h_m)), d = squared distances, h_m = param * median(d_m) / (2 log(N+1)) (no gradient
    through the median) or param^2 with median=False (kernels.py:197-283)."""
    learn = isinstance(param, torch.Tensor) and (param.requires_grad or param.ndim > 0)
    if (_same(Y, X) and X.ndim == 3 and median_ma is None and grad_var is None and not learn
            and not (detach_diag and not readd)):
        # detach_right: K(X, X.detach()) - same values, gradient through the left argument only (weight / function SVGD)
        return ops.transformed_gram_ad(X, ops.GramSpec(ops.GRAM_RBF, _scalar(param), median), left_only=bool(detach_right))
    single = X.ndim == 2
    if single:
        X = X.unsqueeze(0)
        Y = Y.unsqueeze(0) if Y is not None else None
    Yv = X if Y is None else Y
    if detach_right:
        Yv = Yv.detach()
    XY = ops.cross_gram(X, Yv) if (detach_right or not _same(Y, X)) else ops.linear_gram(X, False)
    dists = -2 * XY + X.square().sum(2).unsqueeze(2) + Yv.square().sum(2).unsqueeze(1)
    if detach_diag:
        dists = detach_diagonal(dists, readd=readd)
    B = X.shape[0]
    if isinstance(param, float) or not isinstance(param, torch.Tensor):
        params = float(param) * torch.ones(B, 1, 1, device=X.device, dtype=X.dtype)
    elif param.ndim > 0 and param.shape[0] > 1:
        params = param.view(param.shape[0], 1, 1)
    else:
        params = param.squeeze() * torch.ones(B, 1, 1, device=X.device, dtype=X.dtype)
    if median:
        with torch.no_grad():
            weight = 2.0 * math.log(X.size(1) + 1)
            params = params.clone()
            if median_ma is not None:
                if median_update:
                    rnd = random.randint(0, B - 1)      # one randomly picked member approximates the batch (:257)
                    med = median_ma(dists[rnd].median().detach() / weight, update=True)
                else:
                    med = median_ma.get()
                for i in range(B):
                    params[i] = param * med
            else:
                for i in range(B):
                    params[i] = param * dists[i].median().detach() / weight
    else:
        params = torch.square(params)
    grams = (-dists / (2.0 * params)).exp()
    if single:
        grams = grams.squeeze(0)
    if grad_var is not None:
        return grams, torch.autograd.grad(grams.sum(), grad_var, retain_graph=True)[0]
    return grams


def laplacian_gram(X, Y=None, *args, **kwargs):
    raise NotImplementedError('Currently laplacian kernel is not complete')      # the reference's own words (:301)


MAPS = {
    'linear': linear_gram,
    'log': log_gram,
    'cossim': cossim_gram,
    'hyperenergy': hyperenergy_gram,
    'rbf': rbf_gram,
    'laplacian': laplacian_gram,
}


def learned_gram(X, Y=None, detach_diag=True, readd=True, *args, **kwargs):
    """A feature model (one shared module, or one per member) in front of one of the kernels above (kernels.py:374-422).
    With Y given and a shared model the reference maps X a second time (:413) - kept as it is there."""
    same = Y is None
    if same:
        Y = X
    model = kwargs['model']
    B, N, P = X.shape
    if isinstance(model, (tuple, list)):
        X = torch.stack([m(X[i]) for i, m in enumerate(model)])
    else:
        X = model(X.reshape(B * N, P)).reshape(B, N, -1)
    if same:
        Y = X
    elif isinstance(model, (tuple, list)):
        Y = torch.stack([m(Y[i]) for i, m in enumerate(model)])
    else:
        Y = model(X.reshape(B * N, P)).reshape(B, N, -1)
    method = MAPS[kwargs['learned_kernel']]
    kwargs['detach_diag'] = detach_diag
    kwargs['readd'] = readd
    return method(X, None if same else Y, **kwargs)


def detach_diagonal(gram, readd=True):
    """Diagonal detached from the graph (tiny index op on the caller's tensor); readd=False leaves it zero."""
    eye = torch.eye(gram.shape[1], device=gram.device, dtype=gram.dtype)
    out = gram * (1.0 - eye)
    return out + (gram * eye).detach() if readd else out


def center_gram(gram, ma=None, update_ma=True):
    """One-sided centring: gram - rowmean(gram); with an EMA the subtracted value is the moving average of each member's
    overall Gram mean (kernels.py:335-361)."""
    if ma is None:
        return ops.center_rows(gram)
    mean = ma(gram.mean((1, 2), keepdim=True), update=update_ma)
    if mean.shape[0] != gram.shape[0]:
        mean = mean.mean(0).view(1, -1, 1)
    return gram - mean


def batched_gram(X, Y=None, kernel='linear', detach_diag=True, readd=True, center=True, ma=None, update_ma=True,
                 *args, **kwargs):
    method = learned_gram if kernel == 'learned' else MAPS[kernel]
    gram = method(X, Y, detach_diag=detach_diag, readd=readd, *args, **kwargs).contiguous()
    return center_gram(gram, ma=ma, update_ma=update_ma) if center else gram


def vectorize_grams(gram):
    return gram.reshape(gram.shape[0], gram.shape[1] * gram.shape[2])
