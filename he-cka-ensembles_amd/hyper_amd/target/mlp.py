"""MLP target (hyper/target/mlp.py:7-29): a SequentialModule of batched Linear layers."""
from ..layers.linear import FinalLinear, Linear
from ..layers.module import SequentialModule, register_gen_module


@register_gen_module('mlp')
class MLPModule(SequentialModule):
    def __init__(self, layers, activation='relu', act_last=False, track=True):
        last = len(layers) - 1
        super().__init__(modules=[
            (FinalLinear if layer.get('last', False) else Linear)(
                in_features=layer['in'], out_features=layer['out'], bias=layer.get('bias', True),
                act=layer.get('act', None if (i == last and not act_last) else activation),
                gamma=layer.get('gamma', None if i == 0 else activation), track=layer.get('track', True))
            for i, layer in enumerate(layers)], track=track)

    @staticmethod
    def from_config(config):
        return config
