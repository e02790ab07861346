"""Elementwise activations the target nets use (hyper/net/activation.py: crater :10, leakysoftstep :36,
activation_gamma :48); stock torch elementwise ops."""
import math

import torch

_H = math.sqrt(2.0) / 2.0
_C = _H + 0.008274364699


def crater(x):
    """Variance-preserving GELU-like bump: c * x * (1 + tanh((x + 1) / sqrt(2)))."""
    return _C * x * (1.0 + torch.tanh(_H * (x + 1.0)))


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
