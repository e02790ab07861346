"""Closed-form (no autograd) numpy restatement of the linear-kernel HE / CKA loss and
its analytic gradient, staged exactly as the HIP kernels stage it.

TEST INFRASTRUCTURE ONLY (see oracle/hecka_ref.py for the rules).  It exists to prove
on the CPU, in fp64, that the algebra the kernels rely on equals the reference's
autograd result (hyper/diversity/__init__.py:79-155 + kernels.py:107-127,335-345):

  stage 1  sbar_m = colmean(X_m)                      Xc = X - 1 sbar^T
  stage 2  Kt_m = Xc Xc^T (symmetric)  c_m = Xc sbar  C_m = Kt_m + 1 c_m^T   (= K - rowmean K)
  stage 3  S = <C_i, C_j>                              n_i = sqrt(S_ii)
  stage 4  cos, theta, energy, loss;  G = dLoss/dcos;  A = coefficient matrix with
           dC_m = sum_j A_mj C_j ;  gamma_m = sum_j A_mj c_j
  stage 5  W_m = 2 dC_m + gamma_m 1^T - 1 gamma_m^T,  diag(W_m) = 0 if detach_diag
  stage 6  dX_m = W_m X_m
"""
import numpy as np


def energy_and_slope(theta, s, eps, use_exp):
    """E(theta) and dE/dtheta for the four branches of hyperspherical_energy
    (hyper/diversity/__init__.py:131-144)."""
    if s == 0.0:
        inv = 1.0 / (theta + eps)
        return np.log(inv + eps), (-inv * inv) / (inv + eps)
    if s == 1.0:
        inv = 1.0 / (theta + eps)
        return inv, -inv * inv
    if use_exp:
        e = np.exp(-(s * theta + eps))
        return e, -s * e
    den = theta ** s + eps
    return (1.0 + eps) / den, -(1.0 + eps) * s * theta ** (s - 1.0) / (den * den)


def forward_stats(X):
    """Stages 1-3 for one layer.  X: [M,B,P] -> C [M,B,B], c [M,B], S [M,M]."""
    M, B, _ = X.shape
    sbar = X.mean(axis=1)                                  # [M,P]
    Xc = X - sbar[:, None, :]
    Kt = np.einsum('mbp,mcp->mbc', Xc, Xc)
    c = np.einsum('mbp,mp->mb', Xc, sbar)
    C = Kt + c[:, None, :]
    Cf = C.reshape(M, B * B)
    return C, c, Cf @ Cf.T


def head(S, upstream, head_kind, s, eps, arc_eps, remove_diag, detach_right, abs_vals, use_exp):
    """Stage 4.  Returns (value matrix [M,M] (diag meaningless if remove_diag), A [M,M]).
    `upstream` is dLoss/dvalue_ij as an [M,M] array (zero on the diagonal if remove_diag)."""
    M = S.shape[0]
    n = np.sqrt(np.diag(S))
    d = n + eps
    cos = S / np.outer(d, d)
    if head_kind == 'cka':
        val = cos.copy()
        G = upstream.copy()
    else:
        a = np.abs(cos) if abs_vals else cos
        z = a / (1.0 + arc_eps)
        theta = np.arccos(z)
        val, slope = energy_and_slope(theta, s, eps, use_exp)
        G = upstream * slope * (-1.0 / np.sqrt(1.0 - z * z)) / (1.0 + arc_eps)
        if abs_vals:
            G = G * np.sign(cos)
    if remove_diag:
        G = G * (1.0 - np.eye(M))
    # dLoss/du_i = sum_j Gs_ij u_j ; u_j = v_j / d_j
    Gs = G if detach_right else G + G.T
    A = Gs / np.outer(d, d)
    # normalisation Jacobian: dv_m = g_m/d_m - v_m (v_m.g_m)/(n_m d_m^2)
    vg = (Gs * S / d[None, :]).sum(axis=1)                  # v_m . g_m
    A[np.arange(M), np.arange(M)] += -vg / (n * d * d)
    return val, A


def backward(X, C, c, A, detach_diag):
    """Stages 5-6 for one layer."""
    M, B, _ = X.shape
    dC = np.einsum('mj,jab->mab', A, C)
    gamma = A @ c
    W = 2.0 * dC + gamma[:, :, None] - gamma[:, None, :]
    if detach_diag:
        W = W * (1.0 - np.eye(B))[None]
    return np.einsum('mab,mbp->map', W, X)


def he_loss_and_grad(feats, weights, s=2.0, eps=1e-6, arc_eps=1e-3, use_exp=False,
                     detach_diag=True, detach_right=False, abs_vals=False,
                     head_kind='he', reduction='mean'):
    """Scalar `model_layers` loss (remove_diag, mean|sum) and dX per layer."""
    loss, grads = 0.0, []
    for X, w in zip(feats, weights):
        X = np.asarray(X, dtype=np.float64)
        M = X.shape[0]
        C, c, S = forward_stats(X)
        cnt = M * (M - 1) if reduction == 'mean' else 1
        up = np.full((M, M), w / cnt) * (1.0 - np.eye(M))
        val, A = head(S, up, head_kind, s, abs(eps), abs(arc_eps), True, detach_right, abs_vals, use_exp)
        loss += (val * (1.0 - np.eye(M))).sum() * (w / cnt)
        grads.append(backward(X, C, c, A, detach_diag))
    return loss, grads
