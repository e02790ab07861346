"""The hypernetwork's own small MLPs (hyper/net/models.py:60-188 `build_norm_block`, `build_groupnorm_mlp`;
hyper/net/gpnorm.py:9-31): Linear (spectrally normalised when `coeff` is given) -> GroupNorm -> activation blocks
with row-standardised Gaussian initial weights.  Module names follow the reference (`input`, `center_<i>`,
`group_norm`, `output`; inside a block `linear`, `norm`, `act`) so state-dict keys match."""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.init as init

from . import hip
from .activation import activation_gamma
from .spectral import SpectralLinear


def calculate_mlp_groupnorm_groups(num_features):
    """Group count picked from the feature count (gpnorm.py:18-31)."""
    if num_features > 300:
        for g in (6, 5):
            if num_features % g == 0:
                return g
    if num_features > 200:
        for g in (4, 3):
            if num_features % g == 0:
                return g
    if num_features > 50 and num_features % 2 == 0:
        return 2
    return 1


def build_norm_block(in_features, out_features, bias, act, norm, gamma, norm_groups=None, eps=1e-5, affine=True,
                     coeff=None):
    if coeff is not None:
        linear = SpectralLinear(in_features, out_features, bias=bias, coeff=coeff, n_power_iterations=1)
        weight = linear.weight_orig
    else:
        linear = hip.Linear(in_features, out_features, bias=bias)      # nn.Linear, GPU work in libhecka
        weight = linear.weight
    with torch.no_grad():
        invsq = math.sqrt(1.0 / in_features)
        init.normal_(weight, 0.0, 1.0)
        if bias:
            init.normal_(linear.bias, 0.0, invsq)
        # rows standardised to zero mean and std invsq * gamma
        mean = weight.mean(dim=1, keepdim=True)
        std = weight.std(dim=1, keepdim=True, unbiased=False)
        weight.sub_(mean).mul_((invsq * activation_gamma(gamma)) / (std + eps))
    parts = [('linear', linear)]
    fuse = norm and act is nn.GELU      # GroupNorm -> GELU in one kernel (hecka_groupnorm_*_f32); `act` has no state
    if norm:
        groups = calculate_mlp_groupnorm_groups(out_features) if norm_groups is None else norm_groups
        parts.append(('norm', hip.GroupNorm(num_groups=groups, num_channels=out_features, affine=affine, fused_gelu=fuse)))
    if act is not None and not fuse:
        parts.append(('act', act()))
    return nn.Sequential(OrderedDict(parts))


def mlp_dims_list(in_features, mlp_dims, dim_multiplier):
    """An int `mlp_dims` means that many layers, each `dim_multiplier` times the one before (models.py:129-136)."""
    if isinstance(mlp_dims, (int, float)):
        count = round(mlp_dims)
        dims = [round(in_features * dim_multiplier)]
        for i in range(1, count):
            dims.append(round(dims[i - 1] * dim_multiplier))
        return dims
    return list(mlp_dims)


def build_groupnorm_mlp(in_features, out_features, mlp_dims=3, dim_multiplier=2, bias=False, gamma=1.0, norm_every=2,
                        norm_before_last=False, norm_last=True, norm_last_groups=None, affine_last=True,
                        activation=nn.GELU, out_act=None, coeff=None):
    """`norm_last_groups` is accepted and unused, as in the reference (the output norm is always one group)."""
    dims = mlp_dims_list(in_features, mlp_dims, dim_multiplier)
    layers = OrderedDict()
    layers['input'] = build_norm_block(in_features, dims[0], bias=bias, act=activation, norm=True, gamma=gamma, coeff=coeff)
    for ind in range(1, len(dims)):
        layers['center_%d' % ind] = build_norm_block(dims[ind - 1], dims[ind], bias=bias, gamma=activation,
                                                     act=activation, norm=(ind % norm_every == 0), coeff=coeff)
    assert not (norm_before_last and norm_last), 'Double norming please only set one to true'
    if norm_before_last:
        layers['group_norm'] = hip.GroupNorm(num_groups=calculate_mlp_groupnorm_groups(dims[-1]), num_channels=dims[-1],
                                             affine=True)
    layers['output'] = build_norm_block(dims[-1], out_features, bias=bias, act=out_act, gamma=activation, norm=norm_last,
                                        affine=affine_last, norm_groups=1, coeff=coeff)
    return nn.Sequential(layers)
