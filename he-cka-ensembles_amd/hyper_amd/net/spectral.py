"""Linear layer under soft spectral normalisation (hyper/net/spectral.py:13-107 `SpectralNorm.compute_weight`,
:234-289 `spectral_norm_fc`): the weight used is W / max(1, sigma / coeff) with sigma = u . (W v) from one
power-iteration step per training forward.  u, v are buffers updated in place under no_grad; sigma stays in the
graph (u, v constants), as in the reference.

The reference wraps an nn.Linear with a forward-pre hook; here it is one module with the same state-dict keys
(`weight_orig`, `bias`, `weight_u`, `weight_v`, `weight_sigma`), so checkpoints move both ways.  The power
iteration + sigma run on the current stream (hyper_amd.net.hip: W v, W^T u and the layer's own product in libhecka;
they replay inside a captured step like everything else)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


from . import hip


class SpectralLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True, coeff=1.0, n_power_iterations=1, eps=1e-12):
        super().__init__()
        if n_power_iterations <= 0:
            raise ValueError('Expected n_power_iterations to be positive, but got n_power_iterations=%d' % n_power_iterations)
        self.in_features, self.out_features = in_features, out_features
        self.coeff, self.n_power_iterations, self.eps = coeff, n_power_iterations, eps
        # draws in the reference's order (nn.Linear's own init, then u, then v) so that equal seeds give equal modules
        lin = nn.Linear(in_features, out_features, bias=bias)
        u = F.normalize(lin.weight.new_empty(out_features).normal_(0, 1), dim=0, eps=eps)
        v = F.normalize(lin.weight.new_empty(in_features).normal_(0, 1), dim=0, eps=eps)
        if bias:    # bias before weight_orig: the reference's registration (and state-dict / optimizer) order
            self.bias = nn.Parameter(lin.bias.detach().clone())
        else:
            self.register_parameter('bias', None)
        self.weight_orig = nn.Parameter(lin.weight.detach().clone())
        self.register_buffer('weight_u', u)
        self.register_buffer('weight_v', v)
        self.register_buffer('weight_sigma', torch.ones(1))

    def _sigma(self, do_power_iteration):
        W, u, v = self.weight_orig, self.weight_u, self.weight_v
        if do_power_iteration:
            with torch.no_grad():
                for _ in range(self.n_power_iterations):
                    F.normalize(hip.matvec_t(W.detach(), u), dim=0, eps=self.eps, out=v)
                    F.normalize(hip.matvec(W.detach(), v), dim=0, eps=self.eps, out=u)
                u, v = u.clone(), v.clone()   # the buffers change again next forward; the graph keeps these
        sigma = (u * hip.matvec(W, v)).sum()      # u . (W v) without a BLAS call
        with torch.no_grad():
            self.weight_sigma.copy_(sigma.detach().reshape(1))
        # soft normalisation: only when sigma exceeds coeff.  Written as the reference writes it (a float32 ones(1)
        # against the 0-dim sigma): under torch's promotion rules the factor is float32 even for float64 weights
        return torch.max(torch.ones(1, device=W.device), sigma / self.coeff)

    def compute_weight(self, do_power_iteration):
        """The normalised weight itself (what the reference's hook assigns to `module.weight`)."""
        return self.weight_orig / self._sigma(do_power_iteration)

    def forward(self, x):
        # linear(x, W / f, b) = linear(x, W) / f + b: the scalar goes on the [members, out] result instead of a pass over
        # the weight matrix forward and two backward (the MNIST hypernetwork's output layer is 30840 x 2294)
        y = hip.linear(x, self.weight_orig) / self._sigma(self.training)
        return y if self.bias is None else y + self.bias

    def extra_repr(self):
        return 'in_features=%d, out_features=%d, bias=%s, coeff=%g' % (
            self.in_features, self.out_features, self.bias is not None, self.coeff)


def spectral_norm_fc(in_features, out_features, bias, coeff, n_power_iterations=1, eps=1e-12):
    return SpectralLinear(in_features, out_features, bias=bias, coeff=coeff, n_power_iterations=n_power_iterations, eps=eps)
