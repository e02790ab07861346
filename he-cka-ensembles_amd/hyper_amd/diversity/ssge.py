"""Spectral Stein gradient estimator (hyper/diversity/ssge.py:14-109, after Shi et al. 2018): estimates the score
grad log q(x) at the rows of x from samples xm of q through the leading eigenfunctions of an rbf kernel (Nystrom).
Used by function-space SVGD for the functional prior's score.  The matrices are [members, members] and
[members, samples * classes] - tiny - so this is torch tensor code on the caller's device (one `eigh`, a few products);
the rbf kernel is restated here for two different row sets (the HIP rbf Gram is the self-Gram of the loss path) with the
reference's bandwidth rule: param * median(squared distances) / (2 log(rows + 1)), no gradient through it."""
import math

import torch


def cross_rbf(x, y, param=1.0):
    """exp(-|x_i - y_j|^2 / (2 h)), h from the median heuristic over the [N,M] distance matrix (kernels.py:198-275 with
    the defaults the estimator uses: median=True, detach_diag=True - a value no-op)."""
    d = -2.0 * (x @ y.t()) + (x * x).sum(1)[:, None] + (y * y).sum(1)[None, :]
    with torch.no_grad():
        h = param * d.median() / (2.0 * math.log(x.shape[0] + 1))
    return torch.exp(-d / (2.0 * h))


class SpectralSteinEstimator:
    def __init__(self, eta=None, num_eigs=None, K='rbf', xm=None):
        if K != 'rbf':
            raise NotImplementedError('Only RBF kernel is supported at the moment.')
        self.eta, self.num_eigs = eta, num_eigs
        self.xm = xm
        if xm is not None:
            self.beta, self.eigen_vals, self.eigen_vecs = self.compute_beta(xm)

    def nystrom_method(self, x, eval_points, eigen_vecs, eigen_vals):
        m = float(eval_points.shape[-2])
        return math.sqrt(m) * (cross_rbf(x, eval_points) @ eigen_vecs) / eigen_vals

    def compute_beta(self, xm):
        m = xm.shape[-2]
        with torch.enable_grad():
            xg = xm.detach().requires_grad_(True)
            Kxx = cross_rbf(xg, xg.detach())
            dK = torch.autograd.grad(Kxx.sum(), xg)[0]
        Kxx = Kxx.detach()
        if self.eta is not None:
            Kxx = Kxx + self.eta * torch.eye(m, device=xm.device, dtype=Kxx.dtype)
        eigen_vals, eigen_vecs = torch.linalg.eigh(Kxx)
        if self.num_eigs is not None:
            eigen_vals, eigen_vecs = eigen_vals[:self.num_eigs], eigen_vecs[:, :self.num_eigs]
        beta = -math.sqrt(float(m)) * (eigen_vecs.t() @ (-dK / m)) / eigen_vals[:, None]
        return beta, eigen_vals, eigen_vecs

    def compute_score_gradients(self, x, xm=None):
        if xm is None:
            xm, beta, eigen_vals, eigen_vecs = self.xm, self.beta, self.eigen_vals, self.eigen_vecs
        else:
            beta, eigen_vals, eigen_vecs = self.compute_beta(xm)
        return self.nystrom_method(x, xm, eigen_vecs, eigen_vals) @ beta
