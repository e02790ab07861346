"""Batched target-model building blocks: the GenModule protocol of the reference
(hyper/layers/module.py: registry :12-37, GenModule :41-114, Flatten :208-223, ParametersGroup :226-345,
Activation :348-383, SequentialModule :386-450).  Every module maps (params, x) -> (tracked feature | None, y)
with x = [members, samples, ...]."""
import copy
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init

from ..net import activation as _act

AVAILABLE_GEN_MODULES = {}


def register_gen_module(name):
    def deco(cls):
        AVAILABLE_GEN_MODULES[name] = cls
        return cls
    return deco


def build_gen_module(config):
    config = copy.deepcopy(config)
    if 'name' not in config:
        raise RuntimeError('module config needs a `name` key')
    cls = AVAILABLE_GEN_MODULES[config.pop('name')]
    if hasattr(cls, 'from_config'):
        config = cls.from_config(config)
    return cls(**config)


@register_gen_module('gen_module')
class GenModule(nn.Module):
    def __init__(self, track=True):
        super().__init__()
        self._track = track

    def is_generated(self):
        return False

    def define_generated_modules(self):
        base = OrderedDict()
        if self.is_generated():
            base['self'] = self
        return base

    def single_flat_size(self):
        raise NotImplementedError('GenModule.single_flat_size is not implemented. Please consider using a sub-class')

    def from_flat(self, params):
        return params

    def track_feature(self, feat):
        if self._track == 'detach':
            return feat.detach()
        return feat if self._track else None

    def _forward(self, viewed, x):
        return x

    def forward(self, params, x):
        y = self._forward(self.from_flat(params), x)
        return self.track_feature(y), y

    @staticmethod
    def from_config(config):
        return config


@register_gen_module('flatten')
class Flatten(GenModule):
    def _forward(self, params, x):
        return x.reshape(x.shape[0], x.shape[1], -1)


ACTIVATIONS = {
    'relu': F.relu, 'relu6': F.relu6, 'leaky_relu': F.leaky_relu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh,
    'softmax': F.softmax, 'mish': F.mish, 'elu': F.elu, 'gelu': F.gelu, 'silu': F.silu,
    'crater': _act.crater, 'leakysoftstep': _act.leakysoftstep,
}


@register_gen_module('activation')
class Activation(GenModule):
    def __init__(self, name, track=True):
        super().__init__(track=track)
        if isinstance(name, str) and name not in ACTIVATIONS:
            raise ValueError('invalid activation provided. please select from %s' % list(ACTIVATIONS))
        self._fn = ACTIVATIONS[name] if isinstance(name, str) else name
        self._name = name if isinstance(name, str) else type(name).__name__

    def _forward(self, viewed, x):
        return self._fn(x)


@register_gen_module('parameter_group')
class ParametersGroup(GenModule):
    """A module whose parameters arrive per member as ONE flat row [members, flat]; `from_flat` slices it
    into named views in declaration order (weight first, then bias) without copying."""

    def __init__(self, shapes, shape_type=None, track=True):
        super().__init__(track=track)
        self.shapes = shapes if isinstance(shapes, OrderedDict) else OrderedDict(shapes)
        self.shape_type = shape_type
        if len(self.shapes) == 0:
            raise ValueError('shapes must not be empty')
        for n, s in self.shapes.items():
            if not isinstance(n, str):
                raise ValueError('shape name must be a string')
            if not isinstance(s, (list, tuple)) or len(s) == 0 or any(v == 0 for v in s):
                raise ValueError('shape size must be a non-empty tuple without zero dims')
        self.from_shapes, offset = [], 0
        for name, shape in self.shapes.items():
            flat = int(np.prod(shape))
            self.from_shapes.append((name, tuple(shape), offset, offset + flat))
            offset += flat

    def is_generated(self):
        return True

    def names(self):
        return list(self.shapes.keys())

    def get_shapes(self):
        return self.shapes

    def get_param_flats(self):
        return OrderedDict((n, int(np.prod(s))) for n, s in self.shapes.items())

    def single_flat_size(self):
        return sum(self.get_param_flats().values())

    def gen_empty_params(self, dtype=None, device=None, requires_grad=True):
        return OrderedDict((n, torch.empty(tuple(s), dtype=dtype, device=device, requires_grad=requires_grad))
                           for n, s in self.shapes.items())

    def gen_initialized_params(self, dtype=None, device=None, gain=1.0):
        params = self.gen_empty_params(dtype=dtype, device=device)
        for name in params:
            try:
                init.xavier_normal_(params[name], gain=gain)
            except ValueError:
                init.uniform_(params[name], -0.1, 0.1)
        return params

    def from_flat(self, params):
        m = params.shape[0]
        return OrderedDict((name, params[:, lo:hi].view((m,) + shape)) for name, shape, lo, hi in self.from_shapes)

    def _forward(self, params, x):
        raise NotImplementedError('_forward for ParametersGroup is ill-defined. Please consider using a sub-class')

    def forward(self, params, x):
        y = self._forward(self.from_flat(params['self']), x)
        return self.track_feature(y), y


@register_gen_module('sequential')
class SequentialModule(GenModule, nn.Sequential):
    """nn.Sequential over GenModules that also returns every sub-module's tracked feature."""

    def __init__(self, *args, track=True, modules=None):
        GenModule.__init__(self, track=track)
        nn.Sequential.__init__(self, *(args if len(args) > 0 else (modules or [])))

    def define_generated_modules(self):
        mods = GenModule.define_generated_modules(self)
        for name, module in self._modules.items():
            if isinstance(module, GenModule):
                mods[name] = module.define_generated_modules()
        return mods

    def forward(self, params, x):
        features, current = OrderedDict(), x
        for name, module in self._modules.items():
            sub, current = module(params[name] if isinstance(module, GenModule) else None, current)
            if self._track:
                features[name] = sub
        return features, current

    @staticmethod
    def from_config(config):
        mods = []
        for mod in config['modules']:
            if 'modules' in mod and mod['name'] == 'sequential':
                mods.append(SequentialModule(**SequentialModule.from_config({k: v for k, v in mod.items() if k != 'name'})))
            else:
                mods.append(build_gen_module(mod))
        config['modules'] = mods
        return config
