"""Layer generators of the hypernetwork (hyper/layers/generators/base.py:15-94 registry + BaseLayerGenerator,
:105-201 MLPLayerGenerator; hyper/layers/generators/conv.py:22-105 MLPSampledConvLayerGenerator): map layer codes
[members, codes, code_size] to ONE flat parameter row per member, in the order ParametersGroup.from_flat slices it
(weight first, then bias).  The MLPs are ordinary shared-weight torch modules (net/models.py): the members are their
batch dimension, so they are plain library GEMMs - the batched-weight kernels start where these rows are consumed."""
import copy
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from ..net import models
from .conv import _ConvNd
from .module import ParametersGroup, build_gen_module

AVAILABLE_LAYER_GENERATORS = {}


def register_layer_generator(name):
    def deco(cls):
        AVAILABLE_LAYER_GENERATORS[name] = cls
        return cls
    return deco


def build_layer_generator(config, module):
    config = copy.deepcopy(config)
    if 'name' not in config:
        raise RuntimeError('layer generator config needs a `name` key')
    cls = AVAILABLE_LAYER_GENERATORS[config.pop('name')]
    return cls(definition=module, **cls.from_config(config))


def _null_from_flat(shape):
    raise ValueError('Cannot get shape flat for undefined layer')


class BaseLayerGenerator(nn.Module):
    def __init__(self, definition, input_size):
        super().__init__()
        self.definition, self.input_size = definition, input_size   # the definition registers as a sub-module (no params)

    def get_flat_size(self):
        return self.definition.single_flat_size()

    def get_expected_input_size(self):
        return 1    # codes this layer consumes

    def get_parameter_shapes(self):
        return self.definition.shapes if isinstance(self.definition, ParametersGroup) else OrderedDict()

    def random(self, num, std=1.0, device=None):
        return torch.empty(num, self.get_flat_size(), device=device).normal_(0.0, std)

    def from_flat(self):
        return self.definition.from_flat if isinstance(self.definition, ParametersGroup) else _null_from_flat

    def from_shapes(self):
        return self.definition.from_shapes if isinstance(self.definition, ParametersGroup) else []

    @staticmethod
    def from_config(config):
        if 'definition' in config:
            config['definition'] = build_gen_module(config['definition'])
        return config


@register_layer_generator('mlp_layer_generator')
class MLPLayerGenerator(BaseLayerGenerator):
    """One code -> the whole layer: GroupNorm MLP with a one-group norm on the flat parameter row (`norm_last`)."""

    def __init__(self, definition, input_size, mlp_dims=3, dim_multiplier=2, bias=False, gamma=1.0, norm_last=True,
                 norm_before_last=False, act=None, affine_last=True, sn_coeff=3.0):
        super().__init__(definition, input_size)
        self.generator = models.build_groupnorm_mlp(
            in_features=input_size, out_features=self.get_flat_size(), mlp_dims=mlp_dims, dim_multiplier=dim_multiplier,
            gamma=gamma, bias=bias, norm_last=norm_last, norm_before_last=norm_before_last, activation=nn.GELU,
            affine_last=affine_last, norm_last_groups=1, out_act=None, coeff=sn_coeff)

    def forward(self, codes):
        return self.generator(codes.squeeze(1))


@register_layer_generator('mlp_sampled_conv_generator')
class MLPSampledConvLayerGenerator(BaseLayerGenerator):
    """One code per output channel -> that filter (+ its bias): the MLP runs on [members * out_channels, code_size];
    biases are moved behind all the weights, where from_flat expects them."""

    def __init__(self, definition, input_size, mlp_dims=3, dim_multiplier=2, bias=False, gamma=1.0, norm_before_last=True,
                 sn_coeff=3.0):
        super().__init__(definition, input_size)
        if not isinstance(definition, _ConvNd):
            raise ValueError('Attempted to make a sampled conv filter layer on a non-convolutional definition')
        if definition.transposed:
            raise NotImplementedError('sampled conv generator: transposed convolutions are not supported')
        self.gen_bias = definition.bias
        self.filter_param = self.get_flat_size()
        self.generator = models.build_groupnorm_mlp(
            in_features=input_size, out_features=self.filter_param, mlp_dims=mlp_dims, dim_multiplier=dim_multiplier,
            gamma=gamma, bias=bias, norm_last=False, norm_before_last=norm_before_last, activation=nn.GELU,
            norm_last_groups=1, out_act=None, coeff=sn_coeff)

    def get_flat_size(self):
        taps = (self.definition.in_channels // self.definition.groups) * int(np.prod(self.definition.kernel_size))
        return taps + (1 if self.definition.bias else 0)    # ONE filter

    def get_expected_input_size(self):
        return self.definition.out_channels

    def forward(self, codes):
        members, chans, per = codes.shape[0], self.get_expected_input_size(), self.filter_param
        filters = self.generator(codes.reshape(members * chans, -1)).view(members, chans, per)
        if self.gen_bias:
            weights = filters[:, :, :-1].reshape(members, chans * (per - 1))
            return torch.cat([weights, filters[:, :, -1]], dim=1).contiguous()
        return filters.reshape(members, -1)
