"""ResNet targets for fixed ensembles (hyper/target/resnet.py:22-83 `BasicBlock`, :156-236 `ResNet`, :239-251
`resnet18`): batched Conv2d -> batched BatchNorm2d -> relu blocks with per-member weights; tracked features are the
stem ('conv1'), every `track_every_n`-th block of each stage ('layerK.<i>.block') and the classifier ('linear')."""
from collections import OrderedDict

import torch.nn.functional as F

from ..layers.conv import Conv2d
from ..layers.linear import Linear
from ..layers.module import Activation, GenModule, SequentialModule, build_gen_module, register_gen_module
from ..layers.norm import BatchNorm2d


@register_gen_module('resnet_basic_block')
class BasicBlock(GenModule):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, track=True, act='relu', block_norm=BatchNorm2d, gamma=1.0):
        super().__init__(track=track)
        self.conv1 = Conv2d(in_planes, planes, kernel_size=3, stride=stride, padding=1, bias=False, gamma=gamma)
        self.act = Activation(act)
        self.norm1 = block_norm(planes, track=False)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False, gamma=act)
        self.norm2 = block_norm(planes, track=False)
        self.shortcut = SequentialModule()
        self.use_short = stride != 1 or in_planes != self.expansion * planes
        self.stride = stride
        if self.use_short:
            self.shortcut = SequentialModule(
                Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False, gamma=gamma),
                block_norm(self.expansion * planes))

    def is_generated(self):
        return False

    def define_generated_modules(self):
        mod = super().define_generated_modules()
        mod['conv1'] = self.conv1.define_generated_modules()
        mod['norm1'] = self.norm1.define_generated_modules()
        mod['conv2'] = self.conv2.define_generated_modules()
        mod['norm2'] = self.norm2.define_generated_modules()
        if self.use_short:
            mod['shortcut'] = self.shortcut.define_generated_modules()
        return mod

    def forward(self, params, x):
        feat = OrderedDict()
        if self.act._name == 'relu' and hasattr(self.norm1, 'affine_module'):
            # norm -> relu and norm -> + shortcut -> relu as one launch each (layers/norm.py: BatchedBatchNormAct)
            out = self.norm1(params['norm1'], self.conv1(params['conv1'], x)[1], relu=True)[1]
            short = self.shortcut(params['shortcut'], x)[1] if self.use_short else x
            out = self.norm2(params['norm2'], self.conv2(params['conv2'], out)[1], residual=short, relu=True)[1]
        else:
            out = self.act(None, self.norm1(params['norm1'], self.conv1(params['conv1'], x)[1])[1])[1]
            out = self.norm2(params['norm2'], self.conv2(params['conv2'], out)[1])[1]
            out = out + (self.shortcut(params['shortcut'], x)[1] if self.use_short else x)
            out = self.act(None, out)[1]
        if self._track:
            feat['block'] = out
        return feat, out

    @staticmethod
    def from_config(config):
        return config


@register_gen_module('resnet')
class ResNet(GenModule):
    def __init__(self, block, block_shared=None, block_norm=BatchNorm2d, num_blocks=None, num_classes=10,
                 shared_blocks=0, in_chan=3, planes=(64, 64, 128, 256, 512), act='relu', track_every_n=(2, 2, 1, 1),
                 gamma=1.0):
        super().__init__()
        if shared_blocks > 0:
            raise NotImplementedError('hyper_amd resnet: shared (non-ensemble) blocks are not built '
                                      '(the reference raises for them too, resnet.py:90)')
        self.in_planes = planes[0]
        self._act, self.track_every, self.block_norm = act, track_every_n, block_norm
        self.conv1 = Conv2d(in_chan, planes[0], kernel_size=3, stride=1, padding=1, bias=False, gamma=gamma)
        self.act = Activation(act)
        self.norm1 = block_norm(planes[0], track=False)
        self.num_blocks = num_blocks
        self.layer1 = self._make_layer(block, planes[1], num_blocks[0], 1, track_every_n[0])
        self.layer2 = self._make_layer(block, planes[2], num_blocks[1], 2, track_every_n[1])
        self.layer3 = self._make_layer(block, planes[3], num_blocks[2], 2, track_every_n[2])
        self.layer4 = self._make_layer(block, planes[4], num_blocks[3], 2, track_every_n[3])
        self.linear = Linear(planes[4] * block.expansion, num_classes, act=None, gamma=act, bias=True)

    def _make_layer(self, block, planes, num_blocks, stride, every):
        layers = []
        for ind, st in enumerate([stride] + [1] * (num_blocks - 1)):
            layers.append(block(self.in_planes, planes, st, track=(ind % every == 0), act=self._act,
                                block_norm=self.block_norm, gamma=self._act))
            self.in_planes = planes * block.expansion
        return SequentialModule(*layers)

    def is_generated(self):
        return False

    def define_generated_modules(self):
        mod = super().define_generated_modules()
        mod['conv1'] = self.conv1.define_generated_modules()
        mod['norm1'] = self.norm1.define_generated_modules()
        for name in ('layer1', 'layer2', 'layer3', 'layer4', 'linear'):
            mod[name] = getattr(self, name).define_generated_modules()
        return mod

    def forward(self, params, x):
        feat = OrderedDict()
        if self.act._name == 'relu' and hasattr(self.norm1, 'affine_module'):
            feat['conv1'] = out = self.norm1(params['norm1'], self.conv1(params['conv1'], x)[1], relu=True)[1]
        else:
            feat['conv1'] = out = self.act(None, self.norm1(params['norm1'], self.conv1(params['conv1'], x)[1])[1])[1]
        for name in ('layer1', 'layer2', 'layer3', 'layer4'):
            feat[name], out = getattr(self, name)(params[name], out)
        M, N = out.shape[:2]       # global average pooling
        out = F.avg_pool2d(out.reshape((M * N,) + out.shape[2:]), kernel_size=out.shape[-2:]).view(M, N, -1)
        feat['linear'], out = self.linear(params['linear'], out)
        return feat, out

    @staticmethod
    def from_config(configs):
        configs['block'] = build_gen_module(configs['block'])
        return configs


@register_gen_module('resnet18')
def resnet18(num_classes=10, **kwargs):
    return ResNet(block=BasicBlock, block_norm=BatchNorm2d, num_blocks=[2, 2, 2, 2], num_classes=num_classes,
                  track_every_n=[2, 2, 1, 1], **kwargs)
