"""CPU oracle for the rarely used options of the functional API.

TEST INFRASTRUCTURE ONLY (see oracle/hecka_ref.py: only tests/, smoke() and bench.py's cpu_baseline leg import oracle/).
Plain torch, dtype generic, autograd for the backward.  Parity status: PINNED by tests/golden/functional_f64.npz, made
by the real reference (tests/golden/make_golden.py::run_functional); tests/test_functional_options.py holds this
restatement to it at 1e-12 and the HIP path at 1e-5.  Every function cites the reference lines it follows."""
import math
import random

import torch

from .hecka_ref import detach_diagonal, _offdiag, unit_vector_rows


class EMA(object):
    """hyper/diversity/kernels.py:32-73 (training mode): the first value initialises the shadow, later ones move it by
    (1 - decay)."""

    def __init__(self, decay):
        self.decay, self.shadow = decay, None

    def __call__(self, value, update=True):
        if update:
            if self.shadow is None:
                self.shadow = value.detach().clone()
            else:
                self.shadow = self.shadow - (1.0 - self.decay) * (self.shadow - value.detach())
        return self.shadow

    def get(self):
        return self.shadow


def linear_gram(X, Y=None, detach_diag=True, readd=True):
    """kernels.py:107-127."""
    gram = torch.bmm(X, (X if Y is None else Y).transpose(1, 2))
    return detach_diagonal(gram, readd) if detach_diag else gram


def log_gram(X, Y=None, detach_diag=True, readd=True):
    """kernels.py:130-142."""
    return torch.log(torch.abs(linear_gram(X, Y, detach_diag, readd)) + 1e-6)


def cossim_gram(X, Y=None, detach_diag=True, readd=True):
    """kernels.py:145-169."""
    Yv = X if Y is None else Y
    gram = torch.bmm(X, Yv.transpose(1, 2))
    gram = gram / torch.sqrt(torch.bmm(X.square().sum(2, keepdim=True), Yv.square().sum(2, keepdim=True).transpose(1, 2)))
    return detach_diagonal(gram, readd) if detach_diag else gram


def hyperenergy_gram(X, Y=None, arc_eps=0.005, eps=1e-4, mhe_s=2):
    """kernels.py:172-194."""
    arcs = torch.arccos(cossim_gram(X, Y, False) / (1.0 + arc_eps))
    return (1.0 + eps) / (torch.pow(arcs, mhe_s) + eps)


def rbf_gram(X, Y=None, param=1.0, median=True, detach_diag=True, readd=True, median_ma=None, median_update=True,
             detach_right=False):
    """kernels.py:197-283 with every option: Y, learnable / per-member param, EMA median of one random member."""
    Yv = X if Y is None else Y
    if detach_right:
        Yv = Yv.detach()
    d = -2 * torch.bmm(X, Yv.transpose(1, 2)) + X.square().sum(2).unsqueeze(2) + Yv.square().sum(2).unsqueeze(1)
    if detach_diag:
        d = detach_diagonal(d, readd)
    B = X.shape[0]
    if not isinstance(param, torch.Tensor):
        h = torch.full((B, 1, 1), float(param), dtype=X.dtype)
    elif param.ndim > 0 and param.shape[0] > 1:
        h = param.view(-1, 1, 1)
    else:
        h = param.squeeze() * torch.ones(B, 1, 1, dtype=X.dtype)
    if median:
        with torch.no_grad():
            w = 2.0 * math.log(X.shape[1] + 1)
            if median_ma is not None:
                med = median_ma(d[random.randint(0, B - 1)].median() / w) if median_update else median_ma.get()
                h = (param * med) * torch.ones(B, 1, 1, dtype=X.dtype)
            else:
                h = torch.stack([param * d[i].median() / w for i in range(B)]).view(B, 1, 1)
    else:
        h = torch.square(h)
    return torch.exp(-d / (2.0 * h))


def center_gram(gram, ma=None, update=True):
    """kernels.py:335-361."""
    if ma is None:
        return gram - gram.mean(dim=2, keepdim=True)
    return gram - ma(gram.mean((1, 2), keepdim=True), update)


def half_space_energy(grams, s, eps, arc_eps, remove_diag=True, detach_right=False, use_exp=False):
    """hyper/diversity/__init__.py:79-155 with half_space=True (:98-105): the negated, detached unit vectors are
    appended before the cosine matrix is formed."""
    u = unit_vector_rows(grams, eps)
    u = torch.cat([u, -u.detach()], dim=0)
    cos = u @ (u.t().detach() if detach_right else u.t())
    if remove_diag:
        cos = _offdiag(cos)
    theta = torch.arccos(cos / (1.0 + arc_eps))
    if use_exp:
        return torch.exp(-(s * theta + eps))
    return (1.0 + eps) / (theta ** s + eps)


def normalize_energy(energy, min_he, max_he):
    """hyper/diversity/model_kernels.py:396-405."""
    return (energy - min_he) / (max_he - min_he)
