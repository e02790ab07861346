"""LeNet-5 target (hyper/target/lenet.py:10-56): two batched Conv2d (max-pooled) and three batched Linear layers;
tracked feature keys '0','1','3','4','5' (the Flatten at '2' is untracked)."""
from ..layers.conv import Conv2d
from ..layers.linear import FinalLinear, Linear
from ..layers.module import Flatten, GenModule, SequentialModule, register_gen_module


@register_gen_module('lenet5')
class LeNet(GenModule):
    def __init__(self, in_channels=3, num_classes=10, bias=True, nfnet=False, flatten_dims=256, activation='crater',
                 param_scale='gamma', gamma=1.0, track=True):
        super().__init__(track)
        if nfnet:
            raise NotImplementedError('hyper_amd lenet5: nfnet=True (ScaledWConv2d) is not built')
        self.sequence = SequentialModule(
            Conv2d(in_channels, 6, 5, pooling='max', act=activation, gamma=gamma, bias=bias, param_scale=param_scale),
            Conv2d(6, 16, 5, pooling='max', act=activation, gamma=activation, bias=bias, param_scale=param_scale),
            Flatten(track=False),
            Linear(flatten_dims, 120, bias=bias, act=activation, gamma=activation, param_scale=param_scale),
            Linear(120, 84, bias=bias, act=activation, gamma=activation, param_scale=param_scale),
            FinalLinear(84, num_classes, bias=bias, act=None, gamma=activation, param_scale=param_scale),
            track=track)

    def is_generated(self):
        return False

    def define_generated_modules(self):
        return self.sequence.define_generated_modules()

    def forward(self, params, x):
        return self.sequence(params, x)
