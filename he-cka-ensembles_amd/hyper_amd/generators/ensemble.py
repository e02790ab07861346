"""Fixed ensemble "generator" (hyper/generators/ensemble.py:13-107, hyper/layers/generators/fixed.py:12-75,
forward / feature_split protocol of hyper/generators/base.py:357-491): holds nn.Parameter[size, *shape] per
generated layer, selects members by index and hands every layer ONE flat row per member."""
import copy
from collections import OrderedDict

import torch
import torch.nn as nn

from ..layers.module import GenModule, ParametersGroup, build_gen_module
from ..layers.norm import _NormBuffers
from ..util.collections import DefaultOrderedDict, flatten_keys

AVAILABLE_GENERATORS = {}


def register_generator(name):
    def deco(cls):
        AVAILABLE_GENERATORS[name] = cls
        return cls
    return deco


def build_generator(config):
    config = copy.deepcopy(config)
    if 'name' not in config:
        raise RuntimeError('generator config needs a `name` key')
    cls = AVAILABLE_GENERATORS[config.pop('name')]
    return cls(**cls.from_config(config))


class EnsembleParametersGroupGenerator(nn.Module):
    def __init__(self, definition, fixed_size):
        super().__init__()
        if not isinstance(definition, ParametersGroup):
            raise ValueError('an ensemble generator needs a ParametersGroup (known parameter shapes)')
        if isinstance(fixed_size, float) or fixed_size <= 0:
            raise ValueError('Fixed size must be a non-negative integer')
        self.definition, self.fixed_size = [definition], fixed_size   # list: not a registered sub-module
        members = [definition.gen_initialized_params() for _ in range(fixed_size)]
        self.fixed_parameters = nn.ParameterDict()   # filled one by one: declaration order (weight, bias) matters
        for name in members[0]:
            self.fixed_parameters[name] = nn.Parameter(torch.stack([m[name] for m in members]))
        self.flat_size = sum(int(v[0].numel()) for v in self.fixed_parameters.values())

    def get_expected_input_size(self):
        return self.fixed_size

    def random(self, num, std=1.0, device=None):
        return torch.empty(num, self.flat_size, device=device).normal_(0.0, std)

    def forward(self, codes):
        if getattr(codes, '_hecka_identity', False) and codes.numel() == self.fixed_size:
            # every member, in order (the usual training step): the parameters themselves, no gather in forward and no
            # index_add in backward
            vals = list(self.fixed_parameters.values())
            if len(vals) == 1:      # (a lone weight - every bias-free convolution: a view, not a copy through cat)
                return vals[0].view(self.fixed_size, -1)
            return torch.cat([v.view(self.fixed_size, -1) for v in vals], dim=1)
        idx = codes.long().ravel()
        return torch.cat([v.index_select(0, idx).view(len(idx), -1) for v in self.fixed_parameters.values()], dim=1)


class MemberBuffers:
    """Running-statistics buffers of the selected members (what the reference passes as an OrderedDict of per-member
    tensor lists, hyper/layers/generators/fixed.py:120-131): `bufs[name][i]` is member i's buffer (a view: in-place
    updates reach the ensemble), `gather` / `update` are the vectorised forms the batched BatchNorm uses."""

    def __init__(self, owner, idx):
        self.owner, self.idx = owner, idx

    def keys(self):
        return list(self.owner.names)

    def __getitem__(self, name):
        store = getattr(self.owner, name)
        return [store[i] for i in self.idx.tolist()]

    def _all(self, store):
        return getattr(self.idx, '_hecka_identity', False) and self.idx.numel() == store.shape[0]

    def whole(self, name):
        """Every member, in order: the stacked store itself is the selection (fp32, contiguous, on the GPU)."""
        store = getattr(self.owner, name)
        return self._all(store) and store.is_cuda and store.dtype == torch.float32 and store.is_contiguous()

    def store(self, name):
        return getattr(self.owner, name)

    def gather(self, name):
        store = getattr(self.owner, name)
        return store if self._all(store) else store.index_select(0, self.idx)

    def update(self, name, value, factor):
        store = getattr(self.owner, name)
        if self._all(store):
            store.mul_(1 - factor).add_(value, alpha=factor)
        else:
            store.index_copy_(0, self.idx, (1 - factor) * store.index_select(0, self.idx) + factor * value)


class EnsembleBufferGroupGenerator(nn.Module):
    """Per-member buffers (hyper/layers/generators/fixed.py:78-131).  Stored stacked as [size, *shape] so that one
    tensor op serves all members; state_dict() / load_state_dict() use the reference's per-member keys
    `mod-<member>-<name>`."""

    def __init__(self, definition, fixed_size):
        super().__init__()
        if not isinstance(definition, ParametersGroup):
            raise ValueError('an ensemble buffer generator needs a ParametersGroup (known buffer shapes)')
        if isinstance(fixed_size, float) or fixed_size <= 0:
            raise ValueError('Fixed size must be a non-negative integer')
        self.definition, self.fixed_size = [definition], fixed_size
        members = [definition.gen_initialized_params() for _ in range(fixed_size)]
        self.names = list(members[0].keys())
        for name in self.names:
            self.register_buffer(name, torch.stack([m[name] for m in members]))
        self.flat_size = sum(int(getattr(self, n)[0].numel()) for n in self.names)

    def get_expected_input_size(self):
        return self.fixed_size

    def random(self, num, std=1.0, device=None):
        fresh = EnsembleBufferGroupGenerator(self.definition[0], num)
        return MemberBuffers(fresh.to(device) if device is not None else fresh, torch.arange(num, device=device))

    def forward(self, codes):
        idx = codes.long().ravel()
        if getattr(codes, '_hecka_identity', False):
            idx._hecka_identity = True
        return MemberBuffers(self, idx)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for name in self.names:
            store = getattr(self, name)
            for ind in range(self.fixed_size):
                destination['%smod-%d-%s' % (prefix, ind, name)] = store[ind] if keep_vars else store[ind].detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name in self.names:
            store = getattr(self, name)
            for ind in range(self.fixed_size):
                key = '%smod-%d-%s' % (prefix, ind, name)
                if key in state_dict:
                    with torch.no_grad():
                        store[ind].copy_(state_dict[key])
                elif strict:
                    missing_keys.append(key)


@register_generator('ensemble')
class FixedEnsembleModel(nn.Module):
    def __init__(self, target, ensemble_size):
        super().__init__()
        self.target, self.size = target, ensemble_size
        self.gen_defined = target.define_generated_modules()
        self.layer_generators = nn.ModuleDict()
        self._assign(self.gen_defined, self.layer_generators)

    def _assign(self, definitions, generators):
        for name, mod_def in definitions.items():
            if isinstance(mod_def, dict):
                generators[name] = nn.ModuleDict()
                self._assign(mod_def, generators[name])
            elif isinstance(mod_def, _NormBuffers):
                generators[name] = EnsembleBufferGroupGenerator(mod_def, self.size)
            else:
                generators[name] = EnsembleParametersGroupGenerator(mod_def, self.size)

    def _iter(self, names, generators):
        for name, gen in generators.items():
            if isinstance(gen, nn.ModuleDict):
                yield from self._iter(names + [name], gen)
            else:
                yield names + [name], gen

    def get_target(self):
        return self.target

    def target_parameters(self):
        return list(self.target.parameters())

    def sample_params(self, size, device=None):
        if size is None:
            return torch.arange(0, self.size, device=device, dtype=torch.long)
        if isinstance(size, (list, set)):
            # an explicit member list: the index tensor is kept (no host-to-device copy per step; graph capture needs that)
            key = (tuple(size), str(device))
            cache = self.__dict__.setdefault('_member_id_cache', {})
            if key not in cache:
                if len(cache) > 64:
                    cache.clear()
                cache[key] = torch.tensor(list(size), device=device, dtype=torch.long)
                cache[key]._hecka_identity = list(size) == list(range(self.size))
            return cache[key]
        return torch.randperm(self.size, device=device, dtype=torch.long)[:size]

    def sample_random_params(self, size, device=None):
        return str(size)      # forward_params turns a string into that many N(0,1) members

    def forward_params(self, codes, device=None):
        if isinstance(codes, str):                      # "random members": N(0,1) flat rows
            params = DefaultOrderedDict()
            for names, gen in self._iter([], self.layer_generators):
                params.set_subitems(names, gen.random(int(codes), device=device))
            return params
        if isinstance(codes, int):
            codes = torch.arange(0, codes, device=device, dtype=torch.long)
        elif codes is None:
            codes = torch.arange(0, self.size, device=device, dtype=torch.long)
        assert codes.ndim <= 1, 'Invalid number of code dims. Expecting 1'
        # the range check reads the ids back to the host: once per id tensor, not once per step (the tensors that
        # sample_params keeps for explicit member lists come back here every step; a capture cannot synchronise)
        checked = self.__dict__.setdefault('_checked_codes', {})
        key = (codes.data_ptr(), codes.numel(), codes._version)
        if checked.get(key) is not codes:
            if torch.any(codes < 0) or torch.any(codes >= self.size):
                raise ValueError('Invalid model index in codes. Please only select members from indices 0 to %d' % self.size)
            cached = self.__dict__.get('_member_id_cache', {})
            if any(codes is t for t in cached.values()):
                checked[key] = codes
        idx = codes.reshape(-1).long()
        if getattr(codes, '_hecka_identity', False):
            idx._hecka_identity = True
        params = DefaultOrderedDict()
        for names, gen in self._iter([], self.layer_generators):
            params.set_subitems(names, gen(idx))
        return params

    @staticmethod
    def feature_split(feats, pred=None, ood_N=None, skip_empty=True, detach_feats=None, split_pred_only=False,
                      already_flat=False):
        flat = feats if already_flat else flatten_keys(feats)
        has_ood = ood_N is not None and ood_N > 0
        f_in, f_ood = OrderedDict(), (OrderedDict() if has_ood else None)
        for i, (key, feat) in enumerate(flat.items()):
            if feat is None and skip_empty:
                continue
            if detach_feats is not None and i in detach_feats:
                feat = feat.detach()
            if has_ood and not split_pred_only:
                f_ood[key] = feat[:, -ood_N:]
                feat = feat[:, :-ood_N]
            f_in[key] = feat
        p_in, p_ood = pred, None
        if pred is not None and has_ood:
            p_in, p_ood = pred[:, :-ood_N], pred[:, -ood_N:]
        return f_in, p_in, f_ood, p_ood

    def forward_split(self, params, x, ood_N=0, skip_empty=True, detach_feats=None, split_pred_only=False):
        all_feats, all_pred = self.target(params, x)
        all_feats = flatten_keys(all_feats)
        feats, pred, ood_feats, ood_pred = self.feature_split(all_feats, all_pred, ood_N, skip_empty, detach_feats,
                                                              split_pred_only, already_flat=True)
        return {'feats': all_feats, 'pred': all_pred, 'feats_ind': feats, 'pred_ind': pred,
                'feats_ood': ood_feats, 'pred_ood': ood_pred}

    def forward(self, params, x, sample_params=False, ret_params=False, feature_split=False, **split_args):
        if sample_params:
            params = self.forward_params(self.sample_params(params, device=x.device))
        elif not isinstance(params, OrderedDict):
            params = self.forward_params(params)
        if feature_split:
            res = self.forward_split(params, x, **split_args)
            if ret_params:
                res['params'] = params
            return res
        outs = self.target(params, x)
        return (params, outs) if ret_params else outs

    def unbiased_cka(self, data, num_samples=5, device='cuda', save_path=None, title=None):
        """generators/base.py:224 (see experiments.metrics.unbiased_cka)."""
        from ..experiments.metrics import unbiased_cka
        return unbiased_cka(self, data, num_samples=num_samples, device=device, save_path=save_path, title=title)

    @staticmethod
    def from_config(config):
        if isinstance(config['target'], dict):
            config['target'] = build_gen_module(config['target'])
        return config
