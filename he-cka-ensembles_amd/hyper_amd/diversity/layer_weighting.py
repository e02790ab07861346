"""Per-layer weights of the multi-layer diversity loss — host-side scalars.

Same registry names, ctor kwargs and numeric behaviour as the reference's
hyper/diversity/layer_weighting.py (registry :8-29, base :32-80, tensor :83-140, list :143-172,
linear :175-196, uniform :199-206, exponential :209-230, start_end_linear :233-266)."""
import copy
import math

import torch
import torch.nn as nn

AVAILABLE_WEIGHTING = {}


def register_weighting(name):
    def deco(cls):
        AVAILABLE_WEIGHTING[name] = cls
        return cls
    return deco


def build_weighting(config):
    config = copy.deepcopy(config)
    if 'name' not in config:
        raise RuntimeError('layer_weighting config needs a `name` key')
    cls = AVAILABLE_WEIGHTING[config.pop('name')]
    return cls(**cls.from_config(config))


class LayerWeighting(nn.Module):
    """weight(layer) = f(layer) / norm_ord(f(0..total-1)); the norm is cached per forward."""

    def __init__(self, ord=1):
        super().__init__()
        self.total = None
        self.ord = ord

    def get_weight(self, layer, total, norm=True):
        raise NotImplementedError('Must implement get_weight')

    def _apply_ord(self, n):
        if self.ord is None:
            return n
        if self.ord == 1:
            return abs(n)
        if self.ord >= 2:
            return abs(n) ** self.ord
        raise ValueError

    def total_sum(self, total):
        s = sum(self.get_weight(i, total, norm=False) for i in range(total))
        if self.ord is None or self.ord == 1:
            return s
        if self.ord == 2:
            return math.sqrt(s)
        if self.ord > 2:
            return s ** (1.0 / self.ord)

    def save_total(self, total):
        self.total = self.total_sum(total)

    def save_total_if_needed(self, total):
        if self.total is None:
            self.save_total(total)

    def reset_total(self):
        self.total = None

    def _normed(self, value, total, norm):
        if not norm:
            return value
        self.save_total_if_needed(total)
        return value / self.total

    @staticmethod
    def from_config(config):
        return config


@register_weighting('tensor')
class TensorLayerWeighting(LayerWeighting):
    """Weights held as an nn.Parameter (optionally learnable); normalised by clamp(sum, eps)."""

    def __init__(self, weights, normalize=True, learnable=False, eps=1e-6):
        super().__init__()
        if isinstance(weights, int):
            weights = [1.0 / weights] * weights
        if not isinstance(weights, (torch.Tensor, list)):
            raise ValueError('Weights must be a tensor or a list')
        if isinstance(weights, list):
            weights = torch.tensor(weights, dtype=torch.float32)
        else:
            weights = weights.clone().detach()
        self.weights = nn.Parameter(weights, requires_grad=learnable)
        self.norm = normalize
        self.eps = eps

    def get_weight(self, layer, total, norm=True):
        if layer >= len(self.weights):
            raise ValueError('Layer {} not in weights'.format(layer))
        if self.norm and norm:
            self.save_total_if_needed(total)
            return self.weights[layer] / self.total
        return self.weights[layer]

    def total_sum(self, total):
        return torch.clamp(self.weights.sum(), min=self.eps)

    def reset_total(self):
        # the reference also drops the grad and re-projects the weights here (layer_weighting.py:130-140)
        self.weights.grad = None
        with torch.no_grad():
            if self.norm and self.total is not None:
                self.weights.data = torch.abs(self.weights.data)
                self.weights.data = self.weights.data / self.total_sum(None)
        self.total = None


@register_weighting('list')
class ListLayerWeighting(LayerWeighting):
    def __init__(self, weights, normalize=True, ord=None):
        super().__init__(ord=ord)
        if weights is None:
            raise ValueError('Weights must be a list')
        self.weights = weights
        self.normalize = normalize

    def get_weight(self, layer, total, norm=True):
        if layer >= len(self.weights):
            raise ValueError('Layer {} not in weights'.format(layer))
        return self._normed(self._apply_ord(self.weights[layer]), total, norm and self.normalize)


@register_weighting('linear')
class LinearLayerWeighting(LayerWeighting):
    def __init__(self, start_weight=1.0, weight_increase=1.0, ord=None):
        super().__init__(ord=ord)
        self.start = start_weight
        self.increase = weight_increase

    def get_weight(self, layer, total, norm=True):
        return self._normed(self._apply_ord(self.start + self.increase * layer), total, norm)


@register_weighting('uniform')
class UniformLayerWeighting(LinearLayerWeighting):
    def __init__(self):
        super().__init__(1.0, 0.0)


@register_weighting('exponential')
class ExponentialLayerWeighting(LayerWeighting):
    def __init__(self, start_weight=1.0, weight_increase=1.0, ord=None):
        super().__init__(ord=ord)
        self.start = start_weight
        self.increase = weight_increase

    def get_weight(self, layer, total, norm=True):
        return self._normed(self._apply_ord(self.start * self.increase ** layer), total, norm)


@register_weighting('start_end_linear')
class StartEndLinearLayerWeighting(LayerWeighting):
    """first / last layers pinned, the middle ones linear.  (The reference prints the value on
    every call, layer_weighting.py:264; that debugging print is not reproduced.)"""

    def __init__(self, first_layer, middle_start, middle_increase, last_layer=None, ord=None):
        super().__init__(ord=ord)
        self.first = first_layer
        self.middle_start = middle_start
        self.increase = middle_increase
        self.last = last_layer

    def get_weight(self, layer, total, norm=True):
        if layer == 0:
            value = self.first
        elif layer == total - 1 and self.last is not None:
            value = self.last
        else:
            value = self.middle_start + self.increase * (layer - 1)
        return self._normed(self._apply_ord(value), total, norm)
