"""Elementwise activations the target nets use (hyper/net/activation.py: crater :10, leakysoftstep :36,
activation_gamma :48); stock torch elementwise ops."""
import math

import torch

_H = math.sqrt(2.0) / 2.0
_C = _H + 0.008274364699


class _Crater(torch.autograd.Function):
    """One pass forward, one backward (hecka_crater_f32); keeps x only."""

    @staticmethod
    def forward(ctx, x):
        from .. import _lib
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.lib().hecka_crater_f32(x.data_ptr(), None, y.data_ptr(), x.numel(),
                                               torch.cuda.current_stream(x.device).cuda_stream), 'crater')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().hecka_crater_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(),
                                               torch.cuda.current_stream(x.device).cuda_stream), 'crater (backward)')
        return dx


class _CraterMaxPool2(torch.autograd.Function):
    """maxpool2(crater(x)) over the last two dims without writing the activated tensor (hecka_crater_maxpool2_f32)."""

    @staticmethod
    def forward(ctx, x):
        from .. import _lib
        x = x.contiguous()
        H, W = x.shape[-2:]
        y = torch.empty(x.shape[:-2] + (H // 2, W // 2), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().hecka_crater_maxpool2_f32(x.data_ptr(), None, y.data_ptr(), x.numel() // (H * W), H, W,
                                                        torch.cuda.current_stream(x.device).cuda_stream), 'crater_maxpool2')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib
        x, = ctx.saved_tensors
        H, W = x.shape[-2:]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().hecka_crater_maxpool2_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel() // (H * W), H, W,
                                                        torch.cuda.current_stream(x.device).cuda_stream),
                   'crater_maxpool2 (backward)')
        return dx


def crater(x):
    """Variance-preserving GELU-like bump: c * x * (1 + tanh((x + 1) / sqrt(2))).  fp32 GPU tensors take the one-pass HIP
    kernel; other tensors (fp64 / host-side checks) the five-op torch expression the reference uses."""
    if x.is_cuda and x.dtype == torch.float32 and x.numel() > 0:
        return _Crater.apply(x)
    return _C * x * (1.0 + torch.tanh(_H * (x + 1.0)))


def crater_maxpool2_eligible(x, kwargs):
    """Fused crater -> max_pool2d(kernel_size=2): fp32 GPU tensor [..., H, W] with H even and W a multiple of 4."""
    return (x.is_cuda and x.dtype == torch.float32 and x.numel() > 0 and x.ndim >= 3 and x.shape[-2] % 2 == 0
            and x.shape[-1] % 4 == 0 and dict(kwargs) == {'kernel_size': 2})


def crater_maxpool2(x):
    return _CraterMaxPool2.apply(x)


def softstep(x, k=0.2, clamp=True):
    step = torch.floor(x).detach()
    r = 0.5 - (x - step)
    s = torch.sign(r).detach()
    r = torch.abs(r) * 2.0
    y = step + 0.5 + 0.5 * (s * r / (r * (k - 1.0) - k))
    return torch.clamp(y, 0.0) if clamp else y


def leakysoftstep(x, k=0.55, alpha=0.1):
    y = softstep(x, k=k, clamp=False)
    return torch.where(y > 0, y, alpha * x)


def activation_gamma(act):
    """Variance gain that precedes a layer fed by activation `act`."""
    if act is None:
        return 1.0
    if isinstance(act, (int, float)):
        return float(act)
    if isinstance(act, type):       # a torch.nn activation class names its function
        act = {'ReLU': 'relu', 'LeakyReLU': 'leaky_relu', 'GELU': 'gelu', 'Crater': 'crater'}.get(act.__name__, act)
    if isinstance(act, str):
        return {'crater': 1.0000000001, 'gelu': 1.70092624336, 'leaky_relu': 1.70483162718,
                'relu': math.sqrt(2.0 / (1.0 - (1.0 / math.pi)))}.get(act, 1.0)
    raise ValueError('The provided activation %s is not a valid choice for gamma calculation' % (act,))
