"""Hypernetwork model generators (hyper/generators/base.py:502-679 `LayerCodeModelGenerator`, :682-795
`GeneratedLayerCodeModelGenerator`; hyper/generators/mlp.py:8-44 `MLPLayerModelGenerator`): a latent sample per member
-> code mixer MLP -> one code (conv: one per output channel) per generated layer -> layer generators -> the nested
parameter dict the batched target consumes.  forward / forward_split / feature_split are FixedEnsembleModel's (the
reference shares them through ModelGenerator)."""
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from ..layers.generators import build_layer_generator
from ..layers.module import AVAILABLE_GEN_MODULES, build_gen_module
from ..net import models
from ..util.collections import DefaultOrderedDict, flatten_keys
from .ensemble import FixedEnsembleModel, build_generator, register_generator


class ModelGenerator(nn.Module):
    """What every generator shares (base.py:56-117, :357-499)."""

    def __init__(self, target=None):
        super().__init__()
        self.target = target
        self.gen_defined = self.define_generated_modules()

    def get_target(self):
        return self.target

    def target_parameters(self):
        return self.get_target().parameters()

    def define_generated_modules(self):
        return OrderedDict() if self.target is None else self.target.define_generated_modules()

    def sample_params(self, size, device=None):
        raise NotImplementedError('Sample params not defined for base class model. Use subclass')

    def forward_params(self, x=None):
        raise NotImplementedError('Forward params not defined for the base model generator class. Consider using a sub-class')

    def sample_forward(self, size, x=None):
        return self.forward(self.forward_params(self.sample_params(size, device=x.device)), x)

    feature_split = staticmethod(FixedEnsembleModel.feature_split)
    unbiased_cka = FixedEnsembleModel.unbiased_cka
    forward_split = FixedEnsembleModel.forward_split

    def _run_target(self, params, x):
        return self.target(params, x)

    def forward(self, params, x, sample_params=False, ret_params=False, feature_split=False, **split_args):
        if sample_params:
            if hasattr(self, 'forward_random_params') and isinstance(params, str):
                params = self.forward_random_params(int(params))
            else:
                params = self.forward_params(self.sample_params(params, device=x.device))
        elif not isinstance(params, OrderedDict):
            params = self.forward_params(params)
        if feature_split:
            res = self.forward_split(params, x, **split_args)
            if ret_params:
                res['params'] = params
            return res
        outs = self._run_target(params, x)
        return (params, outs) if ret_params else outs

    @staticmethod
    def from_config(config):
        if isinstance(config['target'], dict):
            config['target'] = build_gen_module(config['target'])
        return config


@register_generator('layer_generator')
class LayerCodeModelGenerator(ModelGenerator):
    def __init__(self, target, code_size, default_generators, specific_generators=None):
        super().__init__(target)
        self.default_gen, self.specific_gen, self.code_size = default_generators, specific_generators, code_size
        self.total_codes = self.total_layers = 0
        self.layer_code_num = OrderedDict()
        self.layer_generators = nn.ModuleDict()
        self._assign(self.gen_defined, self.layer_generators, self.layer_code_num, '')

    def _assign(self, definitions, generators, code_nums, prefix):
        for name, mod_def in definitions.items():
            full = name if prefix == '' else '%s.%s' % (prefix, name)
            if isinstance(mod_def, (dict, nn.ModuleDict)):
                generators[name], code_nums[name] = nn.ModuleDict(), OrderedDict()
                self._assign(mod_def, generators[name], code_nums[name], full)
                continue
            short = full.replace('.self', '')
            if self.specific_gen is not None and full in self.specific_gen:
                generators[name] = self.specific_gen[full](mod_def)
            elif self.specific_gen is not None and short in self.specific_gen:
                generators[name] = self.specific_gen[short](mod_def)
            elif mod_def.__class__ in self.default_gen:
                generators[name] = self.default_gen[mod_def.__class__](mod_def)
            else:
                raise ValueError('the definition class %s for layer %s does not have a respective generator. Available defaults %s'
                                 % (mod_def.__class__.__name__, full, [k.__name__ for k in self.default_gen]))
            code_nums[name] = int(generators[name].get_expected_input_size())
            self.total_codes += code_nums[name]
            self.total_layers += 1

    def get_total_codes(self):
        return self.total_codes

    def get_total_layers(self):
        return self.total_layers

    def _iter(self, names, generators, code_nums):
        for name, gen in generators.items():
            if isinstance(gen, nn.ModuleDict):
                yield from self._iter(names + [name], gen, code_nums[name])
            else:
                yield names + [name], gen, code_nums[name]

    def iter_bfs(self):
        return self._iter([], self.layer_generators, self.layer_code_num)

    def _collect(self, fn):
        out = DefaultOrderedDict()
        for names, gen, _ in self.iter_bfs():
            out.set_subitems(names, fn(gen))
        return out

    def get_parameter_shapes(self):
        return self._collect(lambda g: g.get_parameter_shapes())

    def get_parameter_from_flats(self):
        return self._collect(lambda g: g.from_flat())

    def get_parameter_from_shapes(self):
        return self._collect(lambda g: g.from_shapes())

    def forward_params(self, codes):
        assert codes.ndim == 3, 'Invalid number of code dims. Expecting 3'
        assert codes.shape[1] == self.total_codes, 'Expecting a total of %d codes in axes 1. Got instead %d' % (self.total_codes, codes.shape[1])
        assert codes.shape[2] == self.code_size, 'Expecting a code size of %d in axes 2. Got instead %d' % (self.code_size, codes.shape[2])
        params, offset = DefaultOrderedDict(), 0
        for names, gen, count in self.iter_bfs():
            params.set_subitems(names, gen(codes[:, offset:offset + count, :]))
            offset += count
        return params

    def forward_random_params(self, batch_size, device='cuda'):
        params = DefaultOrderedDict()
        for names, gen, _ in self.iter_bfs():
            params.set_subitems(names, gen.random(batch_size, device=device))
        return params

    @staticmethod
    def from_config(config):
        config = ModelGenerator.from_config(config)
        config['default_generators'] = {AVAILABLE_GEN_MODULES[k]: partial(build_layer_generator, v)
                                        for k, v in config['default_generators'].items()}
        spec = config.get('specific_generators')
        # keyed by layer path ("0.conv2d"); the reference looks the keys up in the module registry here, which fails for
        # any path - a path is what __init__ matches against
        config['specific_generators'] = None if spec is None else {k: partial(build_layer_generator, v) for k, v in spec.items()}
        return config


@register_generator('generated_layer_generator')
class GeneratedLayerCodeModelGenerator(ModelGenerator):
    def __init__(self, latent_size, layer_code_generator, code_generator=None):
        super().__init__(layer_code_generator.get_target())
        self.latent_size, self.layer_gen, self.code_gen = latent_size, layer_code_generator, code_generator

    def set_code_generator(self, code_generator):
        self.code_gen = code_generator

    def get_parameter_shapes(self):
        return self.layer_gen.get_parameter_shapes()

    def get_parameter_from_flats(self):
        return self.layer_gen.get_parameter_from_flats()

    def get_parameter_from_shapes(self):
        return self.layer_gen.get_parameter_from_shapes()

    def forward_layer_params(self, codes):
        if codes.shape[1] != self.layer_gen.total_codes:
            codes = codes.view(codes.shape[0], self.layer_gen.total_codes, self.layer_gen.code_size)
        if codes.shape[2] != self.layer_gen.code_size:
            raise ValueError('Invalid code size passed to forward_layers')
        return self.layer_gen.forward_params(codes)

    def forward_params(self, x, device='cuda'):
        if isinstance(x, str):      # "random members"
            return self.layer_gen.forward_random_params(int(x), device=device)
        if self.code_gen is None:
            raise NotImplementedError('The base class requires a code_generator module. Please set it using set_code_generator')
        return self.forward_layer_params(self.code_gen(x))

    def sample_random_params(self, size, device=None):
        return str(size)

    def _run_target(self, params, x):
        return self.layer_gen.target(params, x)

    def forward(self, params, x, sample_params=False, ret_params=False, feature_split=False, **split_args):
        return super().forward(params, x, sample_params=sample_params or isinstance(params, int), ret_params=ret_params,
                               feature_split=feature_split, **split_args)

    @staticmethod
    def from_config(config, code_gen=None):
        config['layer_code_generator'] = build_generator(config['layer_code_generator'])
        if code_gen is not None:
            config['code_generator'] = code_gen
        return config


@register_generator('mlp_layer_code_generator')
class MLPLayerModelGenerator(GeneratedLayerCodeModelGenerator):
    """Latent [members, latent_size] -> GroupNorm MLP -> all layer codes at once (mlp.py:8-44)."""

    def __init__(self, latent_size, layer_code_generator, mlp_dims=3, dim_multiplier=2, bias=False, gamma=1.0, sn_coeff=1.5):
        super().__init__(latent_size, layer_code_generator, code_generator=None)
        self.set_code_generator(models.build_groupnorm_mlp(
            in_features=latent_size, out_features=layer_code_generator.get_total_codes() * layer_code_generator.code_size,
            mlp_dims=mlp_dims, dim_multiplier=dim_multiplier, gamma=gamma, out_act=None, norm_last=True, activation=nn.GELU,
            norm_last_groups=layer_code_generator.get_total_layers(), bias=bias, coeff=sn_coeff))

    def sample_params(self, size, device=None):
        return torch.randn(size, self.latent_size, device=device)
