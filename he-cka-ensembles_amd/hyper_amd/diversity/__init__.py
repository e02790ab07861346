"""Pairwise heads of the diversity loss: unit-normalised vectorised Grams, pairwise cosine
similarity (CKA) and the arccos hyperspherical energy — same functional API as the reference's
hyper/diversity/__init__.py (unit_vector_rows :6, pairwise_cossim :33, pairwise_cka :59,
hyperspherical_energy :79).

The O(M^2 B^2) work (inner products of the [M, B*B] vectorised Grams and their backward) runs in
libhecka; what is left here is arithmetic on [M,M] matrices, kept as torch ops so that learnable
eps / arc_eps (0-dim Parameters in the reference) get their gradients from autograd."""
import torch

from .. import ops

_TINY = 1e-38


def _inner(grams, detach_right):
    """(S, nsq) for a batch of Grams [M,B,B] or vectors [M,K], differentiable."""
    v = grams.reshape(grams.shape[0], -1) if grams.ndim == 3 else grams
    if v.ndim != 2:
        raise ValueError('Expected matrix of [B, P]')
    return ops.pairwise_inner_ad(v, detach_right)


def cos_from_inner(S, nsq, eps, detach_right=False):
    """cos_ij = <v_i, v_j> / ((|v_i|+eps)(|v_j|+eps)); eps is added to the norm (reference :24-30)."""
    d = torch.sqrt(nsq.clamp_min(_TINY)) + eps
    dr = d.detach() if detach_right else d
    return S / (d.unsqueeze(1) * dr.unsqueeze(0))


def _off_diag(mat):
    return mat[~torch.eye(*mat.shape, dtype=torch.bool, device=mat.device)]


def energy_from_cos(cos, s=0.0, eps=3e-5, arc_eps=3e-4, remove_diag=True, abs_vals=False, use_exp=False):
    """The four energy branches of the reference (:110-144) applied to a cosine matrix."""
    if abs_vals:
        cos = torch.abs(cos)
    if remove_diag:
        cos = _off_diag(cos)
    theta = torch.arccos(cos / (1.0 + arc_eps))
    if s == 0.0 or s == 1.0:
        energy = 1.0 / (theta + eps)
        if s == 0.0:
            energy = torch.log(energy + eps)
    elif use_exp:
        energy = torch.exp(-(s * theta + eps))
    else:
        energy = (1.0 + eps) / (torch.pow(theta, s) + eps)
    return energy


def unit_vector_rows(features, eps=1e-6, vectorize=True):
    if features.ndim == 3 and vectorize:
        features = features.reshape(features.shape[0], -1)
    if features.ndim != 2:
        raise ValueError('Expected matrix of [B, P]')
    _, nsq = _inner(features, False)
    return features / (torch.sqrt(nsq.clamp_min(_TINY)).unsqueeze(1) + eps)


def pairwise_cossim(grams, eps=1e-6, reduction='off_diag', detach_right=False):
    S, nsq = _inner(grams, detach_right)
    cos = cos_from_inner(S, nsq, eps, detach_right)
    return _off_diag(cos) if reduction == 'off_diag' else cos


def pairwise_cka(features, kernel='linear', detach_diag=True, *args, **kwargs):
    from .kernels import batched_gram
    cgrams = batched_gram(features, features, *args, kernel=kernel, detach_diag=detach_diag, center=True, **kwargs)
    return pairwise_cossim(cgrams)


def hyperspherical_energy(features, s=0.0, half_space=False, eps=3e-5, arc_eps=3e-4, offset=0.0, reduction='mean',
                          remove_diag=True, detach_right=False, abs_vals=False, use_exp=False):
    S, nsq = _inner(features, detach_right)
    cos = cos_from_inner(S, nsq, eps, detach_right)
    if half_space:
        # the reference appends the negated, DETACHED unit vectors (:98-105): cos of [u; -u_d] with itself is
        #   [[ u.u^T , -u.u_d^T ], [ -u_d.u^T , u_d.u_d^T ]]   (every right factor detached too under detach_right)
        if detach_right:
            cos_r = cos
        else:
            S_r, nsq_r = _inner(features, True)
            cos_r = cos_from_inner(S_r, nsq_r, eps, True)
        cos_d = cos.detach()
        lower_left = -cos_d if detach_right else -cos_r.t()
        cos = torch.cat([torch.cat([cos, -cos_r], dim=1), torch.cat([lower_left, cos_d], dim=1)], dim=0)
    return energy_from_cos(cos, s, eps, arc_eps, remove_diag, abs_vals, use_exp)
