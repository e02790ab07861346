"""Model kernels: the drop-in seam of the diversity loss.

Same registry (`register_model_kernel` / `build_model_kernel`, YAML `name:` keys `feature`, `cka`,
`he`, `model_layers`), ctor kwargs, forward signatures, `track` side effects and state-dict keys as
the reference's hyper/diversity/model_kernels.py (registry :15-37, FeatureKernel :55-127,
CKAModelKernel :208-256, HEModelKernel :259-414, ModelLayersKernel :417-491).

What differs is underneath: for the linear feature kernel the per-layer chain
    bmm -> detach_diagonal -> center_gram -> vector_norm/divide -> matmul
(and its autograd backward) is ONE fused libhecka op, `ops.centered_gram_pairwise`, that writes the
centred Gram once and returns the [M,M] matrix of Gram inner products; only [M,M]-sized arithmetic
(normalise, arccos, energy, reduction, layer weights) stays in torch, which keeps gradients of
learnable eps / arc_eps / layer weights for free.  `function` and `weight` kernels (SVGD methods)
are outside the hot path and are not built."""
import copy
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops
from ..config import knobs
from ..util.collections import flatten_keys
from . import cos_from_inner, energy_from_cos, hyperspherical_energy, pairwise_cossim
from .kernels import KERN_PARAMS, MAPS, batched_gram, center_gram
from .layer_weighting import LayerWeighting, build_weighting  # noqa: F401

AVAILABLE_KERNELS = {}


def register_model_kernel(name):
    def deco(cls):
        AVAILABLE_KERNELS[name] = cls
        return cls
    return deco


def build_model_kernel(config):
    config = copy.deepcopy(config)
    if 'name' not in config:
        raise RuntimeError('model kernel config needs a `name` key')
    cls = AVAILABLE_KERNELS[config.pop('name')]
    return cls(**cls.from_config(config))


class BaseModelKernel(nn.Module):
    def forward(self, params, feats, pred, track, layer=None):
        pass

    @staticmethod
    def from_config(config):
        return config


def host_scalar(owner, param):
    """float(param) without a device round trip on every call: the value is read back once and cached against
    the parameter's version counter and storage (in-place updates, load_state_dict and .to() all change one of
    them).  A `float(p)` on a device parameter is a blocking copy; four of them per layer per evaluation
    serialised the layer streams behind whatever GEMM grid owned the GPU.
    NOT seen: writes through `param.data` (`p.data.fill_(v)`, `p.data = t` onto the same storage) - `.data` has its own
    version counter.  Update eps / arc_eps / rbf param with `p.copy_()` / `p.fill_()` under no_grad or load_state_dict,
    or call `refresh_host_scalars(kernel)` after a `.data` write."""
    cache = owner.__dict__.setdefault('_host_scalars', {})
    key = (id(param), param._version, param.data_ptr())
    hit = cache.get(id(param))
    if hit is None or hit[0] != key:
        hit = (key, float(param.detach().cpu()))
        cache[id(param)] = hit
    return hit[1]


def refresh_host_scalars(module):
    """Drop the cached host copies of every scalar parameter under `module` (see host_scalar)."""
    for m in module.modules():
        m.__dict__.pop('_host_scalars', None)


# (model_bs, bs, arc_eps, eps) -> (min, max) energies of normalize=True, shared by all kernels of a process (:257)
HE_VALUE_CACHE = {}


@register_model_kernel('feature')
class FeatureKernel(BaseModelKernel):
    """Batched Gram over tracked features [M,B,*].  `params` become fp32 0-dim Parameters under
    `self.params.<name>` exactly as in the reference (state-dict compatible).  Unlike the reference
    (model_kernels.py:78-80) the module-global default dict is NOT mutated: each instance starts from
    a copy of the defaults."""

    def __init__(self, kernel, detach_diag=False, readd_diag=True, params=None, learnable=False, *args, **kwargs):
        super().__init__()
        assert kernel in MAPS, 'Kernel {} not found. See kernels.py for available kernels'.format(kernel)
        self.kernel = kernel
        self.detach_diag = detach_diag
        self.readd = readd_diag
        self.args = args
        self.kwargs = kwargs
        self.dparams = dict(KERN_PARAMS.get(kernel, {}))
        if params is not None:
            self.dparams.update(params)
        self.params = nn.ParameterDict(OrderedDict(
            (k, nn.Parameter(torch.tensor(float(v)), requires_grad=learnable)) for k, v in self.dparams.items()))

    @property
    def fusable(self):
        """True when the Gram -> centre -> pairwise chain can run as the single fused op."""
        return self.gram_spec() is not None

    def gram_spec(self):
        """ops.GramSpec for the fused layer op, or None when this configuration needs the un-fused path."""
        if (self.detach_diag and not self.readd) or self.args or any(p.requires_grad for p in self.params.values()):
            return None
        if self.kernel == 'linear' and not self.kwargs:
            return ops.GramSpec(ops.GRAM_LINEAR)
        if self.kernel == 'cossim' and not self.kwargs:
            return ops.GramSpec(ops.GRAM_COSSIM)
        if self.kernel == 'rbf' and set(self.kwargs) <= {'median'}:
            return ops.GramSpec(ops.GRAM_RBF, host_scalar(self, self.params['param']), self.kwargs.get('median', True))
        return None

    @staticmethod
    def flatten(feats, flat_second=True):
        if feats.ndim < (3 if flat_second else 2):
            raise RuntimeError('Features must be have at least 2/3 dimensions (depending on flat second): '
                               '(model bs, bs, dim1) or (model bs, bs). See model_kernels.py')
        if flat_second:
            return feats.reshape(feats.shape[0], feats.shape[1], -1)
        return feats.reshape(feats.shape[0], -1).unsqueeze(0)

    def batched_kernel(self, feats, flat_second=True, grad_vars=None):
        X = self.flatten(feats, flat_second)
        gram = batched_gram(X, None, self.kernel, self.detach_diag, self.readd, False, None, True,
                            *self.args, **self.params, **self.kwargs)
        if not flat_second:
            gram = gram.squeeze(0)
        if grad_vars is not None:
            return gram, torch.autograd.grad(gram.sum(), grad_vars, retain_graph=True)[0]
        return gram

    def forward(self, params, feats, preds, track, layer=None):
        return self.batched_kernel(feats)


class _PairwiseHead(BaseModelKernel):
    """Shared front of the CKA / HE kernels.

    Fast path (`_spec()` is not None: linear kernel, centred, nothing learnable): the whole layer is one
    `ops.diversity_layer` call.  Otherwise the Gram inner products still come from libhecka and the
    [M,M] head stays in torch so that learnable eps / arc_eps receive gradients."""

    def _learnable(self):
        return any(p.requires_grad for p in self.parameters())

    def _spec(self):
        return None

    def _values(self, feats, track, reduce=None, pre=None):
        """Fast path only: value matrix [M,M] (reduce=None) or the reduced scalar."""
        fk = self.feature_kernel
        out, gram, S, nsq = ops.diversity_layer(fk.flatten(feats), self._spec(), fk.detach_diag, reduce, fk.gram_spec(), pre)
        track['gram'] = gram
        track['gram_inner'] = (S, nsq)   # <C_i, C_j> and ||C_i||^2: lets the `ckas` metric skip its own pass over the Grams
        return out

    def shares_gram_launch(self):
        """This layer's centred linear Gram may come from the all-layers launch (ops.centered_gram_multi)."""
        if self._spec() is None:
            return False
        gs = self.feature_kernel.gram_spec()
        return gs is not None and gs.kind == ops.GRAM_LINEAR

    def reduced(self, feats, track, reduce, pre=None):
        """mean / sum of this layer's kernel values as a 0-dim tensor (what ModelLayersKernel needs)."""
        if self._spec() is not None:
            return self._values(feats, track, reduce, pre)
        vals = self(None, feats, None, track)
        return vals.mean() if reduce == 'mean' else vals.sum()

    def _select(self, vals):
        if self.reduction == 'off_diag':
            M = vals.shape[0]
            return vals.reshape(-1).index_select(0, ops.offdiag_index(M, vals.device))
        return vals

    def _cosine(self, feats, eps, track):
        fk = self.feature_kernel
        if self.center and fk.fusable and fk.kernel == 'linear':
            X = fk.flatten(feats)
            S, nsq, gram = ops.centered_gram_pairwise(X, fk.detach_diag, self.detach_right)
        else:
            gram = fk.batched_kernel(feats)
            if self.center:
                gram = center_gram(gram)
            S, nsq = ops.pairwise_inner_ad(gram.reshape(gram.shape[0], -1), self.detach_right)
        track['gram'] = gram  # read by the `ckas` metric (hyper/experiments/metrics.py:75-91)
        return cos_from_inner(S, nsq, eps, self.detach_right)

    @staticmethod
    def from_config(config):
        config = BaseModelKernel.from_config(config)
        config['feature_kernel'] = build_model_kernel(config['feature_kernel'])
        return config


@register_model_kernel('cka')
class CKAModelKernel(_PairwiseHead):
    def __init__(self, feature_kernel, eps=1e-6, center=True, reduction='off_diag', abs_vals=False,
                 detach_right=False, learnable=False):
        super().__init__()
        self.feature_kernel = feature_kernel
        self.center = center
        self.abs_vals = abs_vals  # stored, never applied - as in the reference (:226, :231-249)
        self.reduction = reduction
        self.detach_right = detach_right
        self.eps = nn.Parameter(torch.tensor(eps), requires_grad=learnable)

    def _spec(self):
        if not (self.center and self.feature_kernel.fusable) or self._learnable():
            return None
        return ops.HeadSpec(ops.HEAD_CKA, eps=host_scalar(self, self.eps), remove_diag=self.reduction == 'off_diag',
                            detach_right=self.detach_right)

    def forward(self, params, feats, preds, track, layer=None):
        if self._spec() is not None:
            return self._select(self._values(feats, track))
        return self._select(self._cosine(feats, self.eps, track))


@register_model_kernel('he')
class HEModelKernel(_PairwiseHead):
    def __init__(self, feature_kernel, he_s=2.0, eps=1e-6, arc_eps=1e-3, center=True, reduction='off_diag',
                 abs_vals=False, detach_right=False, use_exp=False, normalize=False, learnable=False):
        super().__init__()
        self.feature_kernel = feature_kernel
        self.center = center
        self.abs_vals = abs_vals
        self.reduction = reduction
        self.detach_right = detach_right
        self.he_s = he_s
        self.eps = nn.Parameter(torch.tensor(eps), requires_grad=learnable)
        self.arc_eps = nn.Parameter(torch.tensor(arc_eps), requires_grad=learnable)
        self.use_exp = use_exp
        self.normalize_he = normalize

    def get_smoothing(self, detach=False, asfloat=False):
        arc, eps = torch.abs(self.arc_eps), torch.abs(self.eps)
        if detach:
            arc, eps = arc.detach(), eps.detach()
        if asfloat:
            return arc.item(), eps.item()
        return arc, eps

    @torch.enable_grad()
    def precompute(self, model_bs, bs, device='cuda'):
        """Min / max energy of `model_bs` Grams of `bs` samples for normalize=True (model_kernels.py:302-367): the minimum
        from 100 Adam steps on random unit vectors (lr 0.05, x 0.1 every 10 steps), the "maximum" from a tight Gaussian
        cluster far from the origin.  Random start on the device: the values differ from run to run, as in the reference."""
        from . import hyperspherical_energy, unit_vector_rows
        fake = torch.empty(model_bs, bs, bs, requires_grad=True, dtype=torch.float32, device=device)
        torch.nn.init.normal_(fake)
        optim = torch.optim.Adam([fake], lr=0.05)
        sched = torch.optim.lr_scheduler.StepLR(optim, step_size=10, gamma=0.1)
        arc, eps = self.get_smoothing(detach=True, asfloat=True)
        kw = dict(half_space=False, s=self.he_s, arc_eps=arc, eps=eps, use_exp=self.use_exp,
                  remove_diag=self.reduction == 'off_diag', detach_right=self.detach_right, abs_vals=self.abs_vals)
        for _ in range(100):
            optim.zero_grad()
            with torch.no_grad():
                fake.data = unit_vector_rows(fake.data, eps=0.0).reshape(model_bs, bs, bs)
            energy = hyperspherical_energy(fake, **kw).mean()
            energy.backward()
            optim.step()
            sched.step()
        self.min_he_energy = energy.min().item()
        with torch.no_grad():
            torch.nn.init.normal_(fake, mean=1000.0, std=0.1)
            self.max_he_energy = hyperspherical_energy(fake, **kw).max().item()
        if self.normalize_he:
            HE_VALUE_CACHE[(model_bs, bs, arc, eps)] = (self.min_he_energy, self.max_he_energy)
        return self.min_he_energy, self.max_he_energy

    def _normalized(self, energy, model_bs, bs, device):
        """(energy - min) / (max - min) with the pre-solved (cached) extremes (model_kernels.py:396-405)."""
        arc, eps = self.get_smoothing()
        min_he, max_he = HE_VALUE_CACHE.get((model_bs, bs, arc.item(), eps.item()), (None, None))
        if min_he is None or max_he is None:
            min_he, max_he = self.precompute(model_bs, bs, device=device)
        return (energy - min_he) / (max_he - min_he)

    def _spec(self):
        if not (self.center and self.feature_kernel.fusable) or self._learnable() or self.normalize_he:
            return None      # (normalize: the affine map is applied to the un-fused head's values)
        return ops.HeadSpec(ops.HEAD_HE, s=self.he_s, eps=host_scalar(self, self.eps), arc_eps=host_scalar(self, self.arc_eps),
                            use_exp=self.use_exp, abs_vals=self.abs_vals, remove_diag=self.reduction == 'off_diag',
                            detach_right=self.detach_right)

    def forward(self, params, feats, preds, track, layer=None):
        if self._spec() is not None:
            return self._select(self._values(feats, track))
        arc, eps = self.get_smoothing()
        cos = self._cosine(feats, eps, track)
        energy = energy_from_cos(cos, self.he_s, eps, arc, self.reduction == 'off_diag', self.abs_vals, self.use_exp)
        if self.normalize_he:
            gram = track['gram']
            energy = self._normalized(energy, gram.shape[0], gram.shape[1], gram.device)
        return energy


@register_model_kernel('model_layers')
class ModelLayersKernel(BaseModelKernel):
    """sum_l weight_l * reduce(layer_kernel_l(feats_l)); `track['layer_track'][l]['gram']` holds the
    centred Gram of layer l."""

    def __init__(self, layer_kernel, layer_weighting, reduction='mean'):
        super().__init__()
        self.different_kernels = isinstance(layer_kernel, list)
        self.layer_kernel = nn.ModuleList(layer_kernel) if self.different_kernels else layer_kernel
        self.layer_weighting = layer_weighting
        self.reduction = reduction

    def apply_reduction(self, model_grams):
        if self.reduction == 'mean':
            return model_grams.mean()
        if self.reduction == 'sum':
            return model_grams.sum()
        return model_grams

    def _layer_value(self, kern, params, feat, pred, ltrack, pre=None):
        if self.reduction in ('mean', 'sum') and isinstance(kern, _PairwiseHead):
            return kern.reduced(feat, ltrack, self.reduction, pre)   # reduced inside the head kernel
        return self.apply_reduction(kern(params, feat, pred, ltrack))

    def _shared_grams(self, kerns, flist):
        """Centred Grams of ALL layers from one launch (None when some layer is not on the fused linear path, or the
        shapes are not interior): a single layer's Gram grid leaves most of the GPU's second scheduling round empty."""
        if not knobs.multi_gram or self.reduction not in ('mean', 'sum'):
            return None
        if not all(isinstance(k, _PairwiseHead) and k.shares_gram_launch() for k in kerns):
            return None
        Xs = [ops._features(FeatureKernel.flatten(f)) for f in flist]
        if not ops.multi_gram_eligible(Xs) or any(x.requires_grad and x.dtype != f.dtype for x, f in zip(Xs, flist)):
            return None
        with torch.no_grad():
            return ops.centered_gram_multi([x.detach() for x in Xs])

    def _fused_layers(self, kerns, flist, layer_track):
        """The weighted sum over the layers that can share ONE autograd node (ops.DiversityLayers: one Gram launch, one
        head launch, one head_bwd / gamma / bound launch for all of them).  Returns (partial loss, set of layer indices it
        covers), or (None, empty) when fewer than two layers qualify: a layer qualifies when it is on the fused linear
        path with a host-float weight and an interior shape (a net's tiny last layer - 10 logits - must not keep its
        64-channel feature maps from the shared launch)."""
        none = (None, frozenset())
        if not (knobs.fused_layers and knobs.multi_gram):
            return none
        if self.reduction not in ('mean', 'sum'):
            return none
        total = len(flist)
        weights = [self.layer_weighting.get_weight(index, total) for index in range(total)]
        cand = [i for i in range(total) if isinstance(kerns[i], _PairwiseHead) and kerns[i].shares_gram_launch()
                and not isinstance(weights[i], torch.Tensor)]
        Xs = {i: ops._features(FeatureKernel.flatten(flist[i])) for i in cand}
        groups = {}
        for i in cand:      # one launch needs one (members, samples, dtype, detach_diag) and interior feature sizes
            X = Xs[i]
            if ops.multi_gram_eligible([X, X]):
                key = (X.shape[0], X.shape[1], X.dtype, kerns[i].feature_kernel.detach_diag)
                groups.setdefault(key, []).append(i)
        best = max(groups.values(), key=len) if groups else []
        pre = None
        if len(best) < 2 or len(best) > 16 or not ops.multi_gram_eligible([Xs[i] for i in best]):
            # no shared Gram launch (the reference's own batch sizes - 300, 100 - are not interior): the Grams stay one
            # launch per layer, but heads, head_bwd, gamma and the bound are still ONE launch each for all layers - at
            # these sizes the step is a chain of few-microsecond kernels, and every launch removed counts
            loose = {}
            for i in cand:
                X = Xs[i]
                if X.is_cuda and X.dtype == torch.float32 and X.ndim == 3 and X.numel() > 0:
                    loose.setdefault((X.shape[0], X.shape[1], kerns[i].feature_kernel.detach_diag), []).append(i)
            best = max(loose.values(), key=len) if loose else []
            if not knobs.fused_ragged:
                return none
            if len(best) < 2:
                return self._fused_transformed(kerns, flist, layer_track, weights)
            if len(best) > 16:
                return none
            with torch.no_grad():
                X0 = Xs[best[0]]      # one buffer for all the Grams: the batched pairwise launch reads it as [L,M,B,B]
                C_all = torch.empty((len(best), X0.shape[0], X0.shape[1], X0.shape[1]), device=X0.device, dtype=torch.float32)
                pre = [ops.centered_gram(Xs[i].detach().contiguous(), True, out=C_all[k]) for k, i in enumerate(best)]
        M = Xs[best[0]].shape[0]
        specs = [kerns[i]._spec() for i in best]
        scales = []
        for i, spec in zip(best, specs):
            count = (M * (M - 1) if spec.remove_diag else M * M) if self.reduction == 'mean' else 1
            scales.append(float(weights[i]) / count if count else float('nan'))
        res = ops.diversity_layers([Xs[i] for i in best], specs, scales, kerns[best[0]].feature_kernel.detach_diag, pre)
        out, S_all, nsq_all = res[0], res[1], res[2]
        for k, index in enumerate(best):
            layer_track[index]['gram'] = res[3 + k]
            layer_track[index]['gram_inner'] = (S_all[k], nsq_all[k])
        return out, frozenset(best)

    def _fused_transformed(self, kerns, flist, layer_track, weights):
        """cossim / rbf layers (toy: rbf, the MNIST hypernetwork: cossim) in one autograd node: per-layer Gram + transform
        launches, ONE pairwise / head / head_bwd launch for all of them."""
        none = (None, frozenset())
        total = len(flist)
        groups = {}
        for i in range(total):
            k = kerns[i]
            if not isinstance(k, _PairwiseHead) or k._spec() is None or isinstance(weights[i], torch.Tensor):
                continue
            gs = k.feature_kernel.gram_spec()
            f = flist[i]
            if gs is None or gs.kind == ops.GRAM_LINEAR or not f.is_cuda or f.ndim < 3 or f.numel() == 0:
                continue
            groups.setdefault((f.shape[0], f.shape[1]), []).append(i)
        best = max(groups.values(), key=len) if groups else []
        if len(best) < 2 or len(best) > 16:
            return none
        M = flist[best[0]].shape[0]
        specs = [kerns[i]._spec() for i in best]
        scales = []
        for i, spec in zip(best, specs):
            count = (M * (M - 1) if spec.remove_diag else M * M) if self.reduction == 'mean' else 1
            scales.append(float(weights[i]) / count if count else float('nan'))
        res = ops.diversity_layers([FeatureKernel.flatten(flist[i]) for i in best], specs, scales, True, None,
                                   [kerns[i].feature_kernel.gram_spec() for i in best])
        S_all, nsq_all = res[1], res[2]
        for k, index in enumerate(best):
            layer_track[index]['gram'] = res[3 + k]
            layer_track[index]['gram_inner'] = (S_all[k], nsq_all[k])
        return res[0], frozenset(best)

    def forward(self, params, feats, pred, track):
        total = len(feats)
        if total == 0:
            return None
        self.layer_weighting.reset_total()
        self.layer_weighting.save_total(total)
        layer_track = [OrderedDict() for _ in range(total)]
        flist = list(feats.values())
        kerns = [self.layer_kernel[i] if self.different_kernels else self.layer_kernel for i in range(total)]
        fused_out, fused = self._fused_layers(kerns, flist, layer_track) if flist[0].is_cuda else (None, frozenset())
        if len(fused) == total:
            track['layer_track'] = layer_track
            return fused_out
        if fused:       # the remaining layers one by one, their weighted values added to the fused partial sum
            out = fused_out
            for index in range(total):
                if index in fused:
                    continue
                value = self._layer_value(kerns[index], params, flist[index], pred, layer_track[index])
                out = out + self.layer_weighting.get_weight(index, total) * value
            track['layer_track'] = layer_track
            return out
        # layers one after another on the current stream (per-layer streams were measured and lost once all layers' Grams
        # became one launch: DESIGN.md 4)
        pre = self._shared_grams(kerns, flist) if flist[0].is_cuda else None
        values = []
        for index, feat in enumerate(flist):
            values.append(self._layer_value(kerns[index], params, feat, pred, layer_track[index],
                                            pre[index] if pre is not None else None))
        weights = [self.layer_weighting.get_weight(index, total) for index in range(total)]
        if all(v.ndim == 0 for v in values) and not any(isinstance(w, torch.Tensor) for w in weights):
            # host-float weights and scalar layer values: one stack + one dot instead of 2 L tiny kernels (and 2 L more
            # in backward); same sum order as the reference's loop up to fp32 rounding of the partial sums
            key = (tuple(float(w) for w in weights), str(values[0].device), values[0].dtype)
            if getattr(self, '_wcache', (None,))[0] != key:
                self._wcache = (key, torch.tensor(key[0], device=values[0].device, dtype=values[0].dtype))
            out = torch.dot(torch.stack(values), self._wcache[1])
        else:
            out = None
            for index, value in enumerate(values):
                term = weights[index] * value
                out = term if out is None else out + term
        track['layer_track'] = layer_track
        return out

    @staticmethod
    def from_config(config):
        config = BaseModelKernel.from_config(config)
        lk = config['layer_kernel']
        config['layer_kernel'] = [build_model_kernel(k) for k in lk] if isinstance(lk, list) else build_model_kernel(lk)
        config['layer_weighting'] = build_weighting(config['layer_weighting'])
        return config


class FunctionKernel(FeatureKernel):
    """Kernel matrix [M,M] between the members' flattened predictions (model_kernels.py:130-147; function-space SVGD)."""

    def forward(self, params, feats, preds, track, layer=None):
        return self.batched_kernel(preds, flat_second=False)


class WeightKernel(FeatureKernel):
    """Kernel matrix [M,M] between the members' flattened parameters (model_kernels.py:150-205; weight-space SVGD).
    BatchNorm running statistics are not parameters (keys with `.norm_buffer.`)."""

    @staticmethod
    def filter_params(param_dict):
        return OrderedDict((k, v) for k, v in flatten_keys(param_dict).items() if '.norm_buffer.' not in k)

    @staticmethod
    def flatten_weights(params):
        flat, empty = flatten_keys(WeightKernel.filter_params(params), include_empty=True)
        tensors = list(flat.values())
        if len(tensors) == 0:
            raise ValueError('No parameters found in the model')
        members = tensors[0].shape[0]
        return {'params': flat, 'empty': empty, 'flat_params': torch.cat([p.reshape(members, -1) for p in tensors], dim=1),
                'model_bs': members, '__flat__': True}

    def forward(self, params, feats, preds, track, layer=None):
        if isinstance(params, dict) and '__flat__' not in params:
            params = self.flatten_weights(params)
        return self.batched_kernel(params['flat_params'], flat_second=False)


AVAILABLE_KERNELS['weight'] = WeightKernel
AVAILABLE_KERNELS['function'] = FunctionKernel
