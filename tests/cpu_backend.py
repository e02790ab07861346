"""CPU stand-in for hyper_amd.ops used ONLY by the gloo tests of the sharding logic: the five device
stages restated with the oracle's closed-form numpy code (oracle/closed_form.py)."""
import numpy as np
import torch

from oracle import closed_form as CF


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def centered_gram(X, return_xmax=False):
    C, c, _ = CF.forward_stats(X.double().numpy())
    return (_t(C), _t(c), X.abs().amax(dim=(1, 2))) if return_xmax else (_t(C), _t(c))


def pairwise_inner(Ca, Cb=None):
    Cb = Ca if Cb is None else Cb
    return Ca.reshape(Ca.shape[0], -1) @ Cb.reshape(Cb.shape[0], -1).t()


def _head(S, nsq, r0, spec, up):
    """Embed the local rows in a full [M,M] problem (other rows zero upstream) and reuse CF.head."""
    Ma, Mb = S.shape
    F_USE_EXP, F_ABS, F_RM, F_DR = 1, 2, 4, 8
    n = np.sqrt(nsq.numpy())
    d = n + spec.eps
    cos = S.numpy() / np.outer(d[r0:r0 + Ma], d)
    if spec.kind == 0:
        val, slope = cos, np.ones_like(cos)
    else:
        a = np.abs(cos) if spec.flags & F_ABS else cos
        z = a / (1.0 + spec.arc_eps)
        val, de = CF.energy_and_slope(np.arccos(z), spec.s, spec.eps, bool(spec.flags & F_USE_EXP))
        slope = de * (-1.0 / np.sqrt(1.0 - z * z)) / (1.0 + spec.arc_eps)
        if spec.flags & F_ABS:
            slope = slope * np.sign(cos)
    mask = np.ones_like(cos)
    if spec.flags & F_RM:
        mask[np.arange(Ma), r0 + np.arange(Ma)] = 0.0
    return val, slope, mask, n, d


def head_forward(S, nsq, r0, spec, scale, want_values=False):
    val, _, mask, _, _ = _head(S, nsq, r0, spec, None)
    return (_t(val) if want_values else None), torch.tensor(float((val * mask).sum() * scale), dtype=torch.float64)


def head_backward(S, nsq, r0, spec, up_scalar, scale, up_mat=None):
    val, slope, mask, n, d = _head(S, nsq, r0, spec, None)
    Ma, Mb = S.shape
    G = slope * mask * float(up_scalar) * scale
    Gs = G if spec.flags & 8 else 2.0 * G          # uniform upstream: G_ji == G_ij
    A = Gs / np.outer(d[r0:r0 + Ma], d)
    vg = (Gs * S.numpy() / d[None, :]).sum(axis=1)
    A[np.arange(Ma), r0 + np.arange(Ma)] += -vg / (n[r0:r0 + Ma] * d[r0:r0 + Ma] ** 2)
    return _t(A)


def gram_backward(A, C_all, cv_all, X, detach_diag, xmax=None):
    A, C, c, Xn = A.numpy(), C_all.numpy(), cv_all.numpy(), X.double().numpy()
    B = Xn.shape[1]
    dC = np.einsum('mj,jab->mab', A, C)
    gam = A @ c
    W = 2.0 * dC + gam[:, :, None] - gam[:, None, :]
    if detach_diag:
        W = W * (1.0 - np.eye(B))[None]
    return _t(np.einsum('mab,mbp->map', W, Xn)).to(X.dtype)


# the stages of the column-sliced scheme (hyper_amd.parallel._SlicedLayers)
def coef_gamma(A, cvec):
    return A @ cvec


def combine_slice(A, Cs, gamma, B, n_start, detach_diag):
    n = np.arange(n_start, n_start + Cs.shape[1])
    a, b = n // B, n % B
    g = gamma.numpy()
    W = 2.0 * (A.numpy() @ Cs.numpy()) + g[:, a] - g[:, b]
    if detach_diag:
        W[:, a == b] = 0.0
    return _t(W)


def gram_bwd_w(W, X, wmax=None, xmax=None):
    return _t(np.einsum('mab,mbp->map', W.numpy(), X.double().numpy())).to(X.dtype)


# ---- the batched / slice-major stages (ops.centered_gram_multi(slice_rows), gram_bwd_multi, combine_slice_batched, ...):
# offered only under CPU_BACKEND_MULTI=1 so that the gloo tests cover both the per-layer and the batched branches of
# hyper_amd.parallel
import os

SLICED_DTYPES = (torch.float32, torch.float64)
CALLS = {}      # which batched stages ran (the gloo test asserts the branch it means to cover was taken)


def _multi():
    return os.environ.get('CPU_BACKEND_MULTI') == '1'


def multi_gram_eligible(Xs):
    return _multi() and len(Xs) >= 2


def centered_gram_multi(Xs, slice_rows=0, one_buffer=False):
    if one_buffer:      # [G, L, M, rows, B]: every layer's rows for one destination rank contiguous
        CALLS['gram_multi_one_buffer'] = CALLS.get('gram_multi_one_buffer', 0) + 1
        per = []
        for X in Xs:
            C, c, xmax = centered_gram(X, True)
            M, B = C.shape[0], C.shape[1]
            per.append((C.view(M, B // slice_rows, slice_rows, B).permute(1, 0, 2, 3), c, xmax))
        buf = torch.stack([p[0] for p in per], dim=1).contiguous()
        return [(buf[:, li], c, xmax) for li, (_, c, xmax) in enumerate(per)], buf
    CALLS['gram_multi_sliced' if slice_rows else 'gram_multi'] = CALLS.get('gram_multi_sliced' if slice_rows else 'gram_multi', 0) + 1
    out = []
    for X in Xs:
        C, c, xmax = centered_gram(X, True)
        if slice_rows:      # [M,B,B] -> [B/rows, M, rows, B]: the send order of the all-to-all
            M, B = C.shape[0], C.shape[1]
            C = C.view(M, B // slice_rows, slice_rows, B).permute(1, 0, 2, 3).contiguous()
        out.append((C, c, xmax))
    return out


def pairwise_self_batched_eligible(M, K):
    return _multi()


def pairwise_self_batched(C_all):
    return torch.stack([pairwise_inner(C) for C in C_all])


def head_forward_batched(S_all, nsq_all, r0, specs, scales):
    return torch.stack([head_forward(S, n, r0, sp, sc)[1] for S, n, sp, sc in zip(S_all, nsq_all, specs, scales)])


def head_backward_batched(S_all, nsq_all, r0, specs, up, scales):
    return torch.stack([head_backward(S, n, r0, sp, up, sc) for S, n, sp, sc in zip(S_all, nsq_all, specs, scales)])


def coef_gamma_batched(A_all, cv_all):
    return torch.stack([coef_gamma(A, c) for A, c in zip(A_all, cv_all)])


def combine_slice_batched(A_all, Cs_all, gamma_all, B, n_start, detach_diag, nsq_all=None):
    CALLS['combine_batched'] = CALLS.get('combine_batched', 0) + 1
    return torch.stack([combine_slice(A, Cs, g, B, n_start, detach_diag) for A, Cs, g in zip(A_all, Cs_all, gamma_all)])


def wbound_batched(A_all, nsq_all, gamma_all):
    return torch.ones(A_all.shape[:2], dtype=torch.float64)       # (only scales the fp16 split on the GPU; unused here)


def gram_bwd_multi_eligible(Ws, Xs, wmaxs, xmaxs, slice_rows=0):
    return _multi()


def blocked_eligible(M, n, blk_rows):
    return os.environ.get('CPU_BACKEND_BLOCKED') == '1'


def pairwise_self_blocked(Cbuf):
    G, L, Ml, n = Cbuf.shape
    CALLS['pairwise_blocked'] = CALLS.get('pairwise_blocked', 0) + 1
    return pairwise_self_batched(Cbuf.permute(1, 0, 2, 3).reshape(L, G * Ml, n))


def combine_slice_blocked(A_part, Cbuf, gamma_part, l0, B, n_start, detach_diag, nsq_part=None):
    G, Lin, Ml, n = Cbuf.shape
    Lc = A_part.shape[0]
    CALLS['combine_blocked'] = CALLS.get('combine_blocked', 0) + 1
    Cs = Cbuf[:, l0:l0 + Lc].permute(1, 0, 2, 3).reshape(Lc, G * Ml, n)
    Ws = torch.stack([combine_slice(A, C, g, B, n_start, detach_diag) for A, C, g in zip(A_part, Cs, gamma_part)])
    return Ws.view(Lc, G, Ml, n).permute(1, 0, 2, 3).contiguous()


def gram_bwd_multi(Ws, Xs, wmaxs, xmaxs, gate=None, slice_rows=0, slice_members=0):
    CALLS['bwd_multi_sliced' if slice_rows else 'bwd_multi'] = CALLS.get('bwd_multi_sliced' if slice_rows else 'bwd_multi', 0) + 1
    out = []
    for W, X in zip(Ws, Xs):
        M, B = X.shape[0], X.shape[1]
        if slice_rows:      # [B/rows, M, rows, B] as it arrived -> [M,B,B]
            W = W.reshape(B // slice_rows, M, slice_rows, B).permute(1, 0, 2, 3).reshape(M, B, B)
        out.append(gram_bwd_w(W.reshape(M, B, B), X))
    return out


# ---- cossim / rbf layers of the sliced scheme: the member-local transform stages (ops.transformed_gram /
# ops.transformed_gram_bwd_w) restated with torch autograd on the CPU
def _kprime(X, gspec, hband=None):
    G = X @ X.transpose(1, 2)
    g = torch.diagonal(G, dim1=1, dim2=2)
    if gspec.kind == 1:                                  # cossim (hyper/diversity/kernels.py:145-169)
        return G / torch.sqrt(g.unsqueeze(2) * g.unsqueeze(1)), g, hband
    d = -2.0 * G + g.unsqueeze(2) + g.unsqueeze(1)        # rbf (kernels.py:198-286), bandwidth without gradient
    if hband is None:
        with torch.no_grad():
            B = X.shape[1]
            if gspec.median:
                hband = torch.stack([gspec.param * d[m].median() / (2.0 * np.log(B + 1.0)) for m in range(X.shape[0])])
            else:
                hband = torch.full((X.shape[0],), gspec.param ** 2, dtype=X.dtype)
    return torch.exp(-d / (2.0 * hband.view(-1, 1, 1))), g, hband


def transformed_gram(X, gspec, center, out=None):
    with torch.no_grad():
        K, g, hband = _kprime(X.double(), gspec)
        rmean = K.mean(dim=2) if center else torch.zeros(K.shape[:2], dtype=K.dtype)
        C = K - rmean.unsqueeze(2)
    if hband is None:
        hband = torch.zeros(X.shape[0], dtype=torch.float64)
    return C.contiguous(), rmean.contiguous(), g.contiguous(), hband


def transformed_gram_bwd_w(W, C, rmean, gdiag, hband, X, gspec, gate=None):
    # W = dK' + dK'^T with a zero diagonal; K'(X) is symmetric, so <W / 2, K'(X)> has the gradient dK' would give
    CALLS['transform_bwd'] = CALLS.get('transform_bwd', 0) + 1
    with torch.enable_grad():
        Xg = X.double().detach().requires_grad_(True)
        K, _, _ = _kprime(Xg, gspec, hband if gspec.kind == 2 else None)
        (dX,) = torch.autograd.grad((0.5 * W.double().reshape(K.shape) * K).sum(), Xg)
    return dX.to(X.dtype)
