"""Batched Gram builders — functional API of the reference's hyper/diversity/kernels.py
(detach_diagonal :76, linear_gram :107, center_gram :335, batched_gram :425, vectorize_grams :456,
KERN_PARAMS :13, MAPS :364), backed by libhecka.

This is the UN-FUSED entry: each function is its own differentiable op, so Gram, centring and the
pairwise head each make a pass over [M,B,B].  The model kernels (model_kernels.py) use the fused
path (ops.centered_gram_pairwise) instead whenever the configuration allows it.

Built: `linear`, `cossim`, `rbf` (every kernel the shipped YAMLs use); `log`, `hyperenergy`, `laplacian`
and `learned` are used by no config and raise NotImplementedError."""
import torch

from .. import ops

KERN_PARAMS = {
    'linear': {},
    'log': {},
    'cossim': {'eps': 1e-6},
    'hyperenergy': {'arc_eps': 0.005, 'eps': 1e-6, 'mhe_s': 2},
    'rbf': {'param': 1.0},
    'laplacian': {'param': 1.0},
}


def _only_self(Y, X):
    if Y is not None and Y is not X:
        raise NotImplementedError('hyper_amd Grams are built for Y=None (every model kernel of the reference calls them so)')


def linear_gram(X, Y=None, detach_diag=True, readd=True, *args, **kwargs):
    """K_m = X_m X_m^T; with detach_diag the diagonal keeps its value (readd) and gets no gradient."""
    _only_self(Y, X)
    if detach_diag and not readd:
        raise NotImplementedError('readd_diag=False (zeroed Gram diagonal) is not built; no shipped config uses it')
    return ops.linear_gram(X, detach_diag)


def _scalar(v):
    return float(v.detach()) if isinstance(v, torch.Tensor) else float(v)


def cossim_gram(X, Y=None, detach_diag=True, readd=True, *args, **kwargs):
    """K_ab = x_a.x_b / sqrt(|x_a|^2 |x_b|^2); the diagonal is identically 1 and carries no gradient, so
    detach_diag does not change the result.  (`eps` is accepted and ignored, as in the reference.)"""
    _only_self(Y, X)
    if detach_diag and not readd:
        raise NotImplementedError('readd_diag=False (zeroed Gram diagonal) is not built; no shipped config uses it')
    return ops.transformed_gram_ad(X, ops.GramSpec(ops.GRAM_COSSIM))


def rbf_gram(X, Y=None, param=1.0, median=True, detach_diag=True, readd=True, grad_var=None, median_ma=None,
             median_update=True, detach_right=False, *args, **kwargs):
    """K = exp(-d / (2 h_m)), d = squared distances, h_m = param * median(d_m) / (2 log(B+1)) (no gradient
    through the bandwidth) or param^2 with median=False."""
    _only_self(Y, X)
    if median_ma is not None or grad_var is not None:
        raise NotImplementedError('EMA bandwidth / grad_var rbf variants are not built')
    if detach_diag and not readd:
        raise NotImplementedError('readd_diag=False (zeroed Gram diagonal) is not built; no shipped config uses it')
    if isinstance(param, torch.Tensor) and param.requires_grad:
        raise NotImplementedError('a learnable rbf `param` is not built')
    # detach_right: K(X, X.detach()) - same values, gradient through the left argument only (weight / function SVGD)
    return ops.transformed_gram_ad(X, ops.GramSpec(ops.GRAM_RBF, _scalar(param), median), left_only=bool(detach_right))


def _not_built(name):
    def fn(*args, **kwargs):
        raise NotImplementedError('Gram kernel `%s` is not built in hyper_amd yet' % name)
    return fn


MAPS = {
    'linear': linear_gram,
    'log': _not_built('log'),
    'cossim': cossim_gram,
    'hyperenergy': _not_built('hyperenergy'),
    'rbf': rbf_gram,
    'laplacian': _not_built('laplacian'),
}


def detach_diagonal(gram, readd=True):
    """Diagonal detached from the graph (tiny index op on the caller's tensor)."""
    eye = torch.eye(gram.shape[1], device=gram.device, dtype=gram.dtype)
    out = gram * (1.0 - eye)
    return out + (gram * eye).detach() if readd else out


def center_gram(gram, ma=None, update_ma=True):
    """One-sided centring: gram - rowmean(gram) (the EMA variant is unused by the model kernels)."""
    if ma is not None:
        raise NotImplementedError('EMA-smoothed centring is not used by any model kernel and is not built')
    return ops.center_rows(gram)


def batched_gram(X, Y=None, kernel='linear', detach_diag=True, readd=True, center=True, ma=None, update_ma=True,
                 *args, **kwargs):
    if kernel not in MAPS:
        raise KeyError(kernel)
    gram = MAPS[kernel](X, Y, detach_diag=detach_diag, readd=readd, *args, **kwargs).contiguous()
    return center_gram(gram, ma=ma, update_ma=update_ma) if center else gram


def vectorize_grams(gram):
    return gram.reshape(gram.shape[0], gram.shape[1] * gram.shape[2])
