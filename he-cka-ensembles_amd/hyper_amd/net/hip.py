"""HIP building blocks of the hypernetwork's shared-weight MLPs (hyper/net/models.py:60-188,
hyper/net/spectral.py:13-107): Linear over the sampled members through libhecka's grouped GEMM (batch of one weight
matrix), GroupNorm with the following GELU fused, and the matrix-vector products of the spectral power iteration.
GPU float32 tensors take the kernels; CPU / float64 tensors (the host-side state-dict and arithmetic checks) keep the
torch expressions the reference is written in."""
import torch
import torch.nn as nn
import torch.nn.functional as F


from .. import _lib
from ..config import knobs

# A/B handle: HECKA_SAMPLER_TORCH=linear,groupnorm,matvec sends the named pieces back to the torch expressions


def _hip(*ts):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in ts)


def _st(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def linear(x, weight, bias=None):
    """F.linear(x, weight, bias) for a weight shared by all members: one grouped-GEMM launch (forward, dgrad, wgrad)."""
    if not _hip(x, weight, bias) or 'linear' in knobs.sampler_torch:
        return F.linear(x, weight, bias)
    from ..layers.linear import BatchedLinear
    lead = x.shape[:-1]
    y = BatchedLinear.apply(x.reshape(1, -1, x.shape[-1]), weight.unsqueeze(0), None if bias is None else bias.unsqueeze(0), 1.0, 1.0)
    return y.reshape(*lead, weight.shape[0])


class _MatVec(torch.autograd.Function):
    """y = W v (hecka_gemv_f32); dW = g v^T (hecka_ger_f32), dv = W^T g (hecka_gemv_t_f32)."""

    @staticmethod
    def forward(ctx, W, v):
        W, v = W.contiguous(), v.contiguous()
        H, Wd = W.shape
        y = torch.empty((H,), device=W.device, dtype=torch.float32)
        _lib.check(_lib.lib().hecka_gemv_f32(W.data_ptr(), v.data_ptr(), H, Wd, y.data_ptr(), _st(W)), 'gemv')
        ctx.save_for_backward(W, v)
        return y

    @staticmethod
    def backward(ctx, g):
        W, v = ctx.saved_tensors
        H, Wd = W.shape
        L = _lib.lib()
        g = g.contiguous()
        dW = dv = None
        if ctx.needs_input_grad[0]:
            dW = torch.empty_like(W)
            one = torch.ones((1,), device=W.device, dtype=torch.float32)
            _lib.check(L.hecka_ger_f32(one.data_ptr(), g.data_ptr(), v.data_ptr(), H, Wd, dW.data_ptr(), _st(W)), 'ger')
        if ctx.needs_input_grad[1]:
            dv = matvec_t(W, g)
        return dW, dv


def matvec(W, v):
    """torch.mv(W, v)."""
    if not _hip(W, v) or 'matvec' in knobs.sampler_torch:
        return torch.mv(W, v)
    return _MatVec.apply(W, v)


def matvec_t(W, u):
    """torch.mv(W.t(), u) without autograd (the power iteration runs under no_grad)."""
    if not _hip(W, u) or not W.is_contiguous() or 'matvec' in knobs.sampler_torch:
        return torch.mv(W.t(), u)
    L = _lib.lib()
    H, Wd = W.shape
    u = u.contiguous()
    nbytes = L.hecka_gemv_t_workspace_bytes(H, Wd)
    ws = torch.empty((max(nbytes, 4) // 4,), device=W.device, dtype=torch.float32)
    y = torch.empty((Wd,), device=W.device, dtype=torch.float32)
    _lib.check(L.hecka_gemv_t_f32(W.data_ptr(), u.data_ptr(), H, Wd, y.data_ptr(), ws.data_ptr(), nbytes, _st(W)), 'gemv_t')
    return y


class _GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, act):
        x = x.contiguous()
        N, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((N, groups), device=x.device, dtype=torch.float32)
        rstd = torch.empty((N, groups), device=x.device, dtype=torch.float32)
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        _lib.check(_lib.lib().hecka_groupnorm_fwd_f32(x.data_ptr(), N, C, groups, w.data_ptr() if w is not None else None,
                                                     b.data_ptr() if b is not None else None, float(eps), int(act),
                                                     y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _st(x)), 'groupnorm_fwd')
        ctx.save_for_backward(x, w, b, mean, rstd)
        ctx.cfg = (groups, int(act))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mean, rstd = ctx.saved_tensors
        groups, act = ctx.cfg
        N, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        # the affine-gradient kernel runs when dweight is given and fills dbias beside it: a frozen weight with a trainable
        # bias (nn.GroupNorm gives that bias its gradient) gets a scratch dweight that is dropped
        want_w = w is not None and ctx.needs_input_grad[1]
        want_b = b is not None and ctx.needs_input_grad[2]
        dw = torch.empty_like(w) if w is not None and (want_w or want_b) else None
        db = torch.empty_like(b) if want_b and dw is not None else None
        if want_b and dw is None:      # a bias without a weight cannot come out of the fused forward; be loud if it ever does
            raise RuntimeError('group_norm: bias gradient requested without an affine weight')
        p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        _lib.check(_lib.lib().hecka_groupnorm_bwd_f32(dy.data_ptr(), x.data_ptr(), N, C, groups, p(w), p(b), mean.data_ptr(),
                                                     rstd.data_ptr(), act, p(dx), p(dw), p(db), _st(x)), 'groupnorm_bwd')
        return dx, (dw if want_w else None), db, None, None, None


def group_norm(x, groups, weight, bias, eps, gelu=False):
    """F.group_norm on [N, C] rows, optionally followed by F.gelu (erf form), in one kernel each way."""
    if x.ndim != 2 or not _hip(x, weight, bias) or x.shape[0] > 65535 or 'groupnorm' in knobs.sampler_torch:
        y = F.group_norm(x, groups, weight, bias, eps)
        return F.gelu(y) if gelu else y
    return _GroupNormAct.apply(x, weight, bias, groups, eps, 1 if gelu else 0)


class Linear(nn.Linear):
    """nn.Linear (same parameters, state-dict keys and initial values) whose GPU forward / backward is the grouped GEMM."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm with the activation that follows it in a norm block (`fused_gelu`) applied in the same kernel."""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True, fused_gelu=False):
        super().__init__(num_groups, num_channels, eps=eps, affine=affine)
        self.fused_gelu = fused_gelu

    def forward(self, x):
        return group_norm(x, self.num_groups, self.weight, self.bias, self.eps, self.fused_gelu)
