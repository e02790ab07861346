"""Deterministic state for the hypernetwork fixtures.  The shipped MNIST hypernetwork holds ~70 M weights - too many
to commit - so neither its weights nor its gradients are stored: both sides (the real reference in make_golden.py, the
build in the tests) overwrite every state-dict entry with values drawn from a generator seeded by the entry's NAME,
scaled to the mean / std the reference's own initialisation gave that entry (scalars kept in the fixture), and
gradients are compared through a digest (norm, sum, 64 seeded samples).  Test data only - nothing of the reference."""
import zlib

import numpy as np
import torch


def _gen(key, salt=0):
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * salt) % (2 ** 31))


def state_stats(module):
    return {k: (float(v.double().mean()), float(v.double().std(unbiased=False)) if v.numel() > 1 else 0.0)
            for k, v in module.state_dict().items()}


@torch.no_grad()
def fill_state(module, stats):
    """Overwrite every state-dict tensor in place (fp64 draws cast to the tensor's dtype)."""
    for k, v in module.state_dict().items():
        if k.endswith('weight_sigma'):
            v.fill_(1.0)
            continue
        draw = torch.randn(v.numel(), generator=_gen(k), dtype=torch.float64).view(v.shape)
        if k.endswith(('weight_u', 'weight_v')):
            draw = draw / draw.norm()
        else:
            mean, std = stats[k]
            draw = mean + (std if std > 0 else 0.1) * draw
        v.copy_(draw.to(v.dtype))


def grad_digest(key, grad):
    g = grad.detach().double().reshape(-1).cpu()
    idx = torch.randint(0, g.numel(), (64,), generator=_gen(key, 1))
    return np.concatenate([[float(g.norm()), float(g.sum())], g[idx].numpy()])
